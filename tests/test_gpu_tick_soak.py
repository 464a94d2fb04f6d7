"""Soak of cvgs_execute_many on host descriptors: for CVGS_SOAK_SECONDS (default 4), ticks of random shape (2-20 chains x 1-70 crops: both inline
argument blocks and the pinned table ring) alternate over three streams with nothing synchronised inside a burst of 40; a sample of every burst's
tensors is checked against the CPU oracle.  CVGS_SOAK_SECONDS=180 for a long run."""
import os
import time

import numpy as np
import pytest

from cvgpuspeedup_amd import cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_ticks_of_random_shape_on_three_streams(oracle, device, lib):
    import torch
    seconds = float(os.environ.get("CVGS_SOAK_SECONDS", "4"))
    fh, fw = 360, 640
    frames = [H.random_u8((fh, fw, 3), seed=9000 + k) for k in range(6)]
    fts = [torch.from_numpy(f).to(device) for f in frames]
    streams = [torch.cuda.Stream() for _ in range(3)]
    rng = np.random.default_rng(12345)
    t_end = time.time() + seconds
    ticks = checked = 0
    while time.time() < t_end:
        burst, pending = [], []
        for _ in range(40):
            chains, outs, meta = [], [], []
            for m in range(int(rng.integers(2, 21))):
                k = int(rng.integers(0, len(frames)))
                crops = H.random_crops(int(rng.integers(1, 71)), fw, fh, seed=int(rng.integers(1, 1 << 30)), wmax=300, hmax=300)
                out = torch.full((len(crops), 3 * 32 * 48), -777.0, dtype=torch.float32, device=device)
                chains.append(H.k1_chain(cvgs.GpuMat.from_tensor(fts[k], cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (32, 48), 3))
                outs.append(out)
                meta.append((k, crops))
            burst.append((chains, outs, meta))
        # the tensors were filled on torch's current stream; the ticks run on other (non-blocking) streams: the fills must have landed first
        # (without this a fill could overtake its tick: the first version of this test lost a whole tensor to -777 once in ~10 minutes)
        torch.cuda.current_stream().synchronize()
        for chains, outs, meta in burst:
            pending.append((outs, meta, cvgs.executeMany(streams[ticks % 3], chains)))
            ticks += 1
        torch.cuda.synchronize()
        for outs, meta, _ in pending[::8]:
            for out, (k, crops) in list(zip(outs, meta))[::3]:
                ref = np.full((len(crops), 3 * 32 * 48), -777.0, dtype=np.float32)
                oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frames[k], cvgs.CV_8UC3), crops,
                                                     cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), (32, 48), 3)))
                H.assert_bit_exact(out.cpu().numpy(), ref, "soak, tick burst ending at %d" % ticks)
                checked += 1
    assert ticks >= 40 and checked >= 5
    print("soak: %d ticks in %.0f s, %d tensors checked" % (ticks, seconds, checked))


def test_nv12_ticks_of_random_shape_on_three_streams(oracle, device, lib):
    """The same soak on decode-side ticks: crops of NV12 / NV21 surfaces (K4's fused launches: one-row and four-row waves, inline descriptors)."""
    import torch
    from cvgpuspeedup_amd import capi
    seconds = float(os.environ.get("CVGS_SOAK_SECONDS", "4"))
    w, h = 640, 360
    f = cvgs.CV_32FC3
    surfs = [H.random_u8((h + h // 2, w), seed=9100 + k) for k in range(6)]
    sts = [torch.from_numpy(s).to(device) for s in surfs]
    streams = [torch.cuda.Stream() for _ in range(3)]
    rng = np.random.default_rng(54321)

    def chain(k, rects, layout, wrap_s, wrap_o, out):
        m = wrap_s(k)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
        return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (32, 48), capi.YUV_LIMITED, capi.BT709, False, layout=layout),
                cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]),
                cvgs.split(f, wrap_o(out), (32, 48))]

    t_end = time.time() + seconds
    ticks = checked = 0
    while time.time() < t_end:
        burst, pending = [], []
        for _ in range(30):
            layout = capi.YUV_NV12 if rng.integers(2) else capi.YUV_NV21
            chains, outs, meta = [], [], []
            for m in range(int(rng.integers(2, 17))):
                k = int(rng.integers(0, len(surfs)))
                rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in
                         H.random_crops(int(rng.integers(1, 71)), w, h, seed=int(rng.integers(1, 1 << 30)), wmin=4, wmax=300, hmin=4, hmax=300)]
                out = torch.full((len(rects), 3 * 32 * 48), -777.0, dtype=torch.float32, device=device)
                chains.append(chain(k, rects, layout, lambda kk: cvgs.GpuMat.from_tensor(sts[kk], cvgs.CV_8UC1), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), out))
                outs.append(out)
                meta.append((k, rects, layout))
            burst.append((chains, outs, meta))
        torch.cuda.current_stream().synchronize()  # the fills land before the ticks (see above)
        for chains, outs, meta in burst:
            pending.append((outs, meta, cvgs.executeMany(streams[ticks % 3], chains)))
            ticks += 1
        torch.cuda.synchronize()
        for outs, meta, _ in pending[::6]:
            for out, (k, rects, layout) in list(zip(outs, meta))[::3]:
                ref = np.full((len(rects), 3 * 32 * 48), -777.0, dtype=np.float32)
                oracle.execute(cvgs.lower(chain(k, rects, layout, lambda kk: cvgs.GpuMat.from_array(surfs[kk], cvgs.CV_8UC1),
                                                lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
                H.assert_bit_exact(out.cpu().numpy(), ref, "NV12 soak, tick burst ending at %d" % ticks)
                checked += 1
    assert ticks >= 30 and checked >= 5
    print("NV12 soak: %d ticks in %.0f s, %d tensors checked" % (ticks, seconds, checked))
