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
