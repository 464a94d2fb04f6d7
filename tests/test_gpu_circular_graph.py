"""CircularTensor::update inside a HIP graph (CVGS_CIRCULAR_CAPTURABLE).  The reference's update is an ordinary
stream launch (include/cvGPUSpeedup.cuh:612-622), so nothing stops a caller from capturing it; the engine's default path keeps
the ring index on the host and refuses capture.  Capturable handles keep the count on the device: 20 updates captured into ONE
graph must replay as the NEXT 20 updates, every time -- checked slot by slot against the oracle's plain sequence of updates,
for both orders, both plane modes and the mirrored ring, with eager updates before and between the replays."""
import ctypes as C

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _read_device(ptr, nbytes):
    import torch
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3) == 0
    return t.cpu().numpy()


@pytest.mark.parametrize("order,mode,mirrored", [
    (cvgs.NewestFirst, cvgs.Standard, False), (cvgs.OldestFirst, cvgs.Standard, False),
    (cvgs.NewestFirst, cvgs.Transposed, False), (cvgs.OldestFirst, cvgs.Transposed, False),
    (cvgs.NewestFirst, cvgs.Standard, True), (cvgs.OldestFirst, cvgs.Standard, True)])
@pytest.mark.parametrize("push", ["pixel", "resize"])
def test_twenty_captured_updates_replay_as_the_next_twenty(oracle, order, mode, mirrored, push):
    import torch
    dev = torch.device("cuda:0")
    W, H_, B, N = 80, 36, 16, 20
    f = cvgs.CV_32FC3
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, order, mode, W, H_, mirrored=mirrored, capturable=True)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, B, order, mode)
    src_shape = (H_, W, 3) if push == "pixel" else (90, 200, 3)
    pw = [cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3])]
    kind = capi.WRITE_TENSOR_T_SPLIT if mode == cvgs.Transposed else capi.WRITE_TENSOR_SPLIT

    def chain(mat, write):
        if push == "pixel":
            return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [mat], 1), cvgs.convertTo(cvgs.CV_8UC3, f), *pw, write]
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, mat, (W, H_)), *pw, write]

    def gpu_update(stream, frame_t):
        wr = ct.write_splitT(f) if mode == cvgs.Transposed else ct.write_split(f)
        ct.update(stream, *chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), wr))

    def oracle_update(frame):
        oc.update(cvgs.lower(chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), cvgs.WriteIOp(kind, f, 16, W, H_, 0, B))))

    def check(what):
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), what)

    frames = [H.random_u8(src_shape, seed=9000 + i) for i in range(N)]
    frames_t = [torch.from_numpy(x).to(dev) for x in frames]
    extra = [H.random_u8(src_shape, seed=9500 + i) for i in range(5)]
    extra_t = [torch.from_numpy(x).to(dev) for x in extra]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    # three eager updates first: the captured sequence must continue from wherever the tensor stands
    for i in range(3):
        gpu_update(torch.cuda.current_stream(), extra_t[i])
        oracle_update(extra[i])
    check("3 eager updates")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for i in range(N):
            gpu_update(torch.cuda.current_stream(), frames_t[i])
    torch.cuda.synchronize()
    check("capture itself must not run anything")
    g.replay()
    for i in range(N):
        oracle_update(frames[i])
    check("first replay = updates 4..23")
    assert ct.updates() == 3 + N
    gpu_update(torch.cuda.current_stream(), extra_t[3])  # an eager update between the replays
    oracle_update(extra[3])
    check("eager update between the replays")
    g.replay()
    for i in range(N):
        oracle_update(frames[i])
    check("second replay = the NEXT twenty updates")
    assert ct.updates() == 4 + 2 * N
    ct.release()


def test_default_handles_still_refuse_capture_and_say_how():
    import torch
    dev = torch.device("cuda:0")
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, 3, cvgs.NewestFirst, cvgs.Standard, 32, 16)
    frame = torch.zeros((16, 32, 3), dtype=torch.uint8, device=dev)
    f = cvgs.CV_32FC3
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            with pytest.raises(capi.CvgsError, match="CVGS_CIRCULAR_CAPTURABLE"):
                ct.update(torch.cuda.current_stream(), cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3), cvgs.convertTo(cvgs.CV_8UC3, f), ct.write_split(f))
            frame.add_(1)
        finally:
            g.capture_end()
    torch.cuda.synchronize()
    ct.release()
