"""The device-side descriptor queue (cvgs_queue_*): every batch a resident server grid produces must be bit-identical to the
CPU oracle -- the same contract as cvgs_execute (tests/test_gpu_k1.py), whose call shape the queue keeps (one submit per frame,
reference include/cvGPUSpeedup.cuh:464-473).  Covered: variable crops, aspect-ratio padding, unused planes, C4, ragged target
sizes, many batches in flight (ring wrap-around), a server that retires and is restarted, a source buffer REWRITTEN between
two submits (the server outlives kernel boundaries, so nothing invalidates its caches for it), a consumer stream waiting on a
ticket, and chains the server must refuse."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def torch_dev():
    import torch
    return torch, torch.device("cuda:0")


def oracle_out(oracle, frame_np, crops, batch_out, dst, cn, **kw):
    ref = np.full((batch_out, cn * dst[0] * dst[1]), -777.0, dtype=np.float32)
    h_src = cvgs.GpuMat.from_array(frame_np, cvgs.make_type(cvgs.CV_8U, cn))
    oracle.execute(cvgs.lower(H.k1_chain(h_src, crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, cn, **kw)))
    return ref


def gpu_chain(torch, dev, frame_t, crops, batch_out, dst, cn, **kw):
    out_t = torch.full((batch_out, cn * dst[0] * dst[1]), -777.0, dtype=torch.float32, device=dev)
    ops = H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.make_type(cvgs.CV_8U, cn)), crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), dst, cn, **kw)
    return out_t, ops


@pytest.mark.parametrize("cn,dst,kw", [
    (3, (64, 128), {}),
    (3, (64, 128), {"swap": False}),
    (4, (64, 128), {}),
    (3, (64, 128), {"ar": cvgs.PRESERVE_AR, "background": [128.0, 128.0, 128.0, 0.0]}),
    (3, (64, 128), {"used": 5, "background": [7.0, 8.0, 9.0, 0.0]}),
    (3, (100, 37), {}),      # ragged: 2 column tiles, the second 36 wide; 37 rows = 2 full tasks + 5 rows
    (4, (30, 50), {"ar": cvgs.PRESERVE_AR_LEFT, "background": [1.0, 2.0, 3.0, 4.0]}),
    (3, (256, 16), {}),      # 4 column tiles, one task row
])
def test_queue_batch_matches_the_oracle(oracle, torch_dev, cn, dst, kw):
    torch, dev = torch_dev
    frame = H.random_u8((1080, 1920, cn), seed=3)
    crops = H.random_crops(9, 1920, 1080, wmax=400, hmax=500, seed=11)
    q = cvgs.Queue()
    try:
        out_t, ops = gpu_chain(torch, dev, torch.from_numpy(frame).to(dev), crops, 9, dst, cn, **kw)
        torch.cuda.synchronize()
        q.wait(q.submit(*ops))
        torch.cuda.synchronize()
        H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, 9, dst, cn, **kw), "queue batch %s %s" % (dst, kw))
    finally:
        q.destroy()


def test_queue_many_batches_in_flight_wrap_the_ring(oracle, torch_dev):
    """200 batches through an 8-slot ring without waiting in between: distinct frames / crop lists / tensors, every one checked."""
    torch, dev = torch_dev
    q = cvgs.Queue(depth=8)
    try:
        frames = [H.random_u8((720, 1280, 3), seed=100 + i) for i in range(5)]
        frames_t = [torch.from_numpy(f).to(dev) for f in frames]
        jobs = []
        for i in range(200):
            crops = H.random_crops(1 + i % 50, 1280, 720, wmax=300, hmax=400, seed=1000 + i)
            out_t, ops = gpu_chain(torch, dev, frames_t[i % 5], crops, len(crops), (64, 128), 3)
            jobs.append((i, crops, out_t, cvgs.lower(ops)))
        torch.cuda.synchronize()
        last = None
        for _, _, _, lowered in jobs:
            last = q.submit_lowered(lowered)
        q.wait(last)
        torch.cuda.synchronize()
        st = q.stats()
        assert st["completed"] == 200 and st["error"] == 0, st
        for i, crops, out_t, _ in jobs:
            H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frames[i % 5], crops, len(crops), (64, 128), 3), "batch %d" % i)
    finally:
        q.destroy()


def test_queue_server_retires_and_restarts(oracle, torch_dev):
    import time
    torch, dev = torch_dev
    q = cvgs.Queue(idle_us=50.0)
    try:
        frame = H.random_u8((480, 640, 3), seed=5)
        frame_t = torch.from_numpy(frame).to(dev)
        for rnd in range(4):
            crops = H.random_crops(12, 640, 480, wmax=200, hmax=300, seed=50 + rnd)
            out_t, ops = gpu_chain(torch, dev, frame_t, crops, 12, (64, 128), 3)
            torch.cuda.synchronize()
            q.wait(q.submit(*ops))
            H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, 12, (64, 128), 3), "round %d" % rnd)
            time.sleep(0.01)  # far beyond idle_us: the server has retired, the next submit launches a new one
        st = q.stats()
        # one launch per round; a host thread descheduled for > idle_us between the launch and its tail write costs one more
        assert 4 <= st["server_launches"] <= 8 and st["error"] == 0, st
    finally:
        q.destroy()


def test_queue_sees_a_source_buffer_rewritten_between_submits(oracle, torch_dev):
    """The same frame buffer gets new pixels while the server stays alive: every batch must be computed from the pixels that were
    in the buffer when it was submitted (ONE server launch serves all six: nothing invalidates its caches between them)."""
    torch, dev = torch_dev
    crops = H.random_crops(20, 1280, 720, wmax=300, hmax=400, seed=9)
    frames = [H.random_u8((720, 1280, 3), seed=200 + rnd) for rnd in range(6)]
    refs = [oracle_out(oracle, f, crops, 20, (64, 128), 3) for f in frames]
    staged = [torch.from_numpy(f).to(dev) for f in frames]
    q = cvgs.Queue(idle_us=200000.0)  # 200 ms: the server survives the host-side work between the submits
    try:
        frame_t = torch.zeros((720, 1280, 3), dtype=torch.uint8, device=dev)
        out_t, ops = gpu_chain(torch, dev, frame_t, crops, 20, (64, 128), 3)
        lowered = cvgs.lower(ops)
        got = [torch.empty_like(out_t) for _ in range(6)]  # allocated up front: a hipMalloc inside the loop would wait for the server
        for rnd in range(6):
            frame_t.copy_(staged[rnd])
            torch.cuda.current_stream().synchronize()  # NOT a device-wide synchronize: that one waits for the server to retire
            q.wait(q.submit_lowered(lowered))
            got[rnd].copy_(out_t)
        torch.cuda.current_stream().synchronize()
        assert q.stats()["error"] == 0, q.stats()
        for rnd in range(6):
            H.assert_bit_exact(got[rnd].cpu().numpy(), refs[rnd], "rewrite %d" % rnd)
    finally:
        q.destroy()


def test_queue_stream_wait_orders_a_consumer_kernel(oracle, torch_dev):
    torch, dev = torch_dev
    q = cvgs.Queue()
    try:
        frame = H.random_u8((720, 1280, 3), seed=21)
        crops = H.random_crops(30, 1280, 720, wmax=300, hmax=400, seed=22)
        out_t, ops = gpu_chain(torch, dev, torch.from_numpy(frame).to(dev), crops, 30, (64, 128), 3)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        t = q.submit(*ops)
        q.stream_wait(t, s)
        with torch.cuda.stream(s):
            copy = out_t.clone()  # a consumer kernel enqueued behind the ticket
        s.synchronize()
        H.assert_bit_exact(copy.cpu().numpy(), oracle_out(oracle, frame, crops, 30, (64, 128), 3), "consumer behind stream_wait")
    finally:
        q.destroy()


def test_queue_refuses_what_the_server_does_not_take(torch_dev):
    torch, dev = torch_dev
    q = cvgs.Queue()
    try:
        frame_t = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
        out_t = torch.zeros((2, 3 * 64 * 128), dtype=torch.float32, device=dev)
        ops = H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), H.fixed_crops(2), cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1))
        ops.insert(-1, cvgs.add(cvgs.CV_32FC3, [1.0, 2.0, 3.0]))  # a fourth arithmetic stage: not the server's program
        with pytest.raises(capi.CvgsError):
            q.submit(*ops)
        frame32 = torch.zeros((480, 640, 3), dtype=torch.float32, device=dev)  # a CV_32F source: cvgs_execute's business
        ops32f = H.k1_chain(cvgs.GpuMat.from_tensor(frame32, cvgs.CV_32FC3), H.fixed_crops(2), cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), src_depth=cvgs.CV_32F)
        with pytest.raises(capi.CvgsError):
            q.submit(*ops32f)
        # a source whose rows span 4 GB or more (the workers address rows with 32-bit byte offsets): refused before anything is read
        huge = cvgs.GpuMat(4, 640, cvgs.CV_8UC3, frame_t.data_ptr(), 1 << 30, owner=frame_t)
        ops_huge = H.k1_chain(huge, [(0, 0, 640, 4)], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1))
        with pytest.raises(capi.CvgsError):
            q.submit(*ops_huge)
        # the queue is still usable afterwards
        out32, ops32 = gpu_chain(torch, dev, frame_t, H.fixed_crops(2), 2, (64, 128), 3)
        q.wait(q.submit(*ops32))
    finally:
        q.destroy()


def test_queue_staging_path_without_host_writes_to_device_memory(oracle, torch_dev):
    """flags bit 0: never write device memory from the host -- slots travel through pinned host copies and a staging kernel
    (what the queue falls back to where the PCIe BAR does not expose device memory)."""
    torch, dev = torch_dev
    q = cvgs.Queue(flags=1)
    try:
        assert q.stats()["host_writes_device_memory"] is False
        frame = H.random_u8((720, 1280, 3), seed=31)
        frame_t = torch.from_numpy(frame).to(dev)
        jobs = []
        for i in range(40):
            crops = H.random_crops(1 + i % 20, 1280, 720, wmax=300, hmax=400, seed=300 + i)
            out_t, ops = gpu_chain(torch, dev, frame_t, crops, len(crops), (64, 128), 3)
            jobs.append((crops, out_t, cvgs.lower(ops)))
        torch.cuda.synchronize()
        tickets = [q.submit_lowered(j[2]) for j in jobs]
        for t in tickets:
            q.wait(t)
        for i, (crops, out_t, _) in enumerate(jobs):
            H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, len(crops), (64, 128), 3), "staged batch %d" % i)
    finally:
        q.destroy()


def test_queue_tiny_batches_on_a_deep_ring(oracle, torch_dev):
    """600 batches of 1-3 crops through a 256-slot ring: a worker's consecutive tasks lie hundreds of batches apart, so the
    64-entry index window has to step through the ring; small targets (one 4-row task per plane) and out-of-order completion."""
    torch, dev = torch_dev
    q = cvgs.Queue(depth=256)
    try:
        frame = H.random_u8((480, 640, 3), seed=41)
        frame_t = torch.from_numpy(frame).to(dev)
        jobs = []
        for i in range(600):
            crops = H.random_crops(1 + i % 3, 640, 480, wmax=200, hmax=300, seed=4000 + i)
            dst = (64, 128) if i % 2 else (48, 20)
            out_t, ops = gpu_chain(torch, dev, frame_t, crops, len(crops), dst, 3)
            jobs.append((crops, dst, out_t, cvgs.lower(ops)))
        torch.cuda.synchronize()
        last = None
        for j in jobs:
            last = q.submit_lowered(j[3])
        for t in range(last + 1):
            q.wait(t)
        st = q.stats()
        assert st["completed"] == 600 and st["error"] == 0, st
        for i, (crops, dst, out_t, _) in enumerate(jobs):
            if i % 7 == 0 or i > 590:
                H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, len(crops), dst, 3), "tiny batch %d" % i)
    finally:
        q.destroy()


def test_queue_two_queues_side_by_side(oracle, torch_dev):
    """Two queues on one device (two server grids share the chip): both produce the oracle's bits."""
    torch, dev = torch_dev
    qa, qb = cvgs.Queue(), cvgs.Queue()
    try:
        frame = H.random_u8((720, 1280, 3), seed=51)
        frame_t = torch.from_numpy(frame).to(dev)
        jobs = []
        for i in range(30):
            crops = H.random_crops(10 + i, 1280, 720, wmax=300, hmax=400, seed=500 + i)
            out_t, ops = gpu_chain(torch, dev, frame_t, crops, len(crops), (64, 128), 3)
            jobs.append((crops, out_t, cvgs.lower(ops)))
        torch.cuda.synchronize()
        tickets = [(qa if i % 2 == 0 else qb).submit_lowered(j[2]) for i, j in enumerate(jobs)]
        for i, t in enumerate(tickets):
            (qa if i % 2 == 0 else qb).wait(t)
        for i, (crops, out_t, _) in enumerate(jobs):
            H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, len(crops), (64, 128), 3), "queue %s batch %d" % ("ab"[i % 2], i))
    finally:
        qa.destroy()
        qb.destroy()


# ---- the queue's second kind: crops of NV12 / NV21 decoder surfaces (K4's shape) ----------------------------------------------
def _nv12_ops(luma_mats, out_mat, dst, range_, prim, layout, swap=True, ar=None, background=None, used=None):
    f = cvgs.CV_32FC3
    rd = cvgs.read_nv12(luma_mats, dst, range_, prim, False, layout=layout)
    if ar is not None:
        rd.ar = ar
    if background is not None:
        rd.background = cvgs._scalar(background)
    if used is not None:
        rd.used_planes = used
    ops = [rd]
    if swap:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f))
    return ops + [cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                  cvgs.split(f, out_mat, dst)]


def _nv12_case(oracle, torch, dev, q, surf, w, h, crops, dst, **kw):
    """one batch: crops (x, y, w, h; even) of the surface `surf` ((h + h/2) x w bytes) on the queue, on cvgs_execute and on the oracle"""
    n = len(crops)
    surf_t = torch.from_numpy(surf).to(dev)
    outs = {}
    for name in ("queue", "execute"):
        out_t = torch.full((n, 3 * dst[0] * dst[1]), -777.0, dtype=torch.float32, device=dev)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf_t.data_ptr(), w, owner=surf_t)
        ops = _nv12_ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), dst, **kw)
        torch.cuda.synchronize()
        if name == "queue":
            q.wait(q.submit(*ops))
        else:
            cvgs.executeOperations(torch.cuda.current_stream(), *ops)
        torch.cuda.synchronize()
        outs[name] = out_t.cpu().numpy()
    ref = np.full((n, 3 * dst[0] * dst[1]), -777.0, dtype=np.float32)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf.ctypes.data, w, owner=surf)
    oracle.execute(cvgs.lower(_nv12_ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, **kw)))
    H.assert_bit_exact(outs["queue"], ref, "NV12 queue batch vs the oracle %s" % (kw,))
    H.assert_bit_exact(outs["queue"], outs["execute"], "NV12 queue batch vs cvgs_execute")


def _even_crops(n, w, h, seed, wmin=8, hmin=8):
    return [tuple(v & ~1 for v in c) for c in H.random_crops(n, w, h, seed=seed, wmin=wmin + 1, wmax=w // 2, hmin=hmin + 1, hmax=h // 2)]


@pytest.mark.parametrize("dst,kw", [
    ((64, 128), dict(range_=capi.YUV_FULL, prim=capi.BT709, layout=capi.YUV_NV12)),
    ((64, 128), dict(range_=capi.YUV_LIMITED, prim=capi.BT601, layout=capi.YUV_NV21)),
    ((64, 128), dict(range_=capi.YUV_LIMITED, prim=capi.BT2020, layout=capi.YUV_NV12, swap=False)),
    ((100, 37), dict(range_=capi.YUV_FULL, prim=capi.BT601, layout=capi.YUV_NV12)),       # ragged: 2 column tiles, 37 rows
    ((64, 64), dict(range_=capi.YUV_LIMITED, prim=capi.BT709, layout=capi.YUV_NV12, ar=cvgs.PRESERVE_AR, background=[114.0, 100.5, 7.25])),
    ((96, 64), dict(range_=capi.YUV_FULL, prim=capi.BT709, layout=capi.YUV_NV21, ar=cvgs.PRESERVE_AR_LEFT, background=[1.0, 2.0, 3.0], used=4)),
    ((256, 16), dict(range_=capi.YUV_FULL, prim=capi.BT709, layout=capi.YUV_NV12)),
])
def test_queue_nv12_batch_matches_the_oracle(oracle, torch_dev, dst, kw):
    torch, dev = torch_dev
    w, h = 1280, 720
    surf = H.random_u8((h + h // 2, w), seed=21)
    crops = _even_crops(7, w, h, seed=31) + [(0, 0, w, h), (w - 4, h - 2, 4, 2)]  # the whole surface, a 4 x 2 corner
    q = cvgs.Queue()
    try:
        _nv12_case(oracle, torch, dev, q, surf, w, h, crops, dst, **kw)
    finally:
        q.destroy()


def test_queue_nv12_whole_6k_surface_cfg3(oracle, torch_dev):
    """BASELINE cfg #3 through the queue: a 6K NV12 surface -> BGR float -> 1280 x 720 -> normalize -> split, frame after frame."""
    torch, dev = torch_dev
    w, h = 6144, 3456
    q = cvgs.Queue()
    try:
        for seed in (41, 42):
            surf = H.random_u8((h + h // 2, w), seed=seed)
            _nv12_case(oracle, torch, dev, q, surf, w, h, [(0, 0, w, h)], (1280, 720), range_=capi.YUV_FULL, prim=capi.BT709, layout=capi.YUV_NV12)
        assert q.stats()["error"] == 0
    finally:
        q.destroy()


def test_queue_nv12_many_batches_in_flight(oracle, torch_dev):
    torch, dev = torch_dev
    w, h = 1920, 1080
    dst = (64, 128)
    surfs = [H.random_u8((h + h // 2, w), seed=50 + i) for i in range(3)]
    surf_ts = [torch.from_numpy(s).to(dev) for s in surfs]
    q = cvgs.Queue(depth=8)
    try:
        jobs = []
        for i in range(40):
            crops = _even_crops(10, w, h, seed=300 + i)
            out_t = torch.full((10, 3 * dst[0] * dst[1]), -777.0, dtype=torch.float32, device=dev)
            st = surf_ts[i % 3]
            luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, st.data_ptr(), w, owner=st)
            ops = _nv12_ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), dst, capi.YUV_LIMITED, capi.BT709, capi.YUV_NV12)
            jobs.append((i, crops, out_t, q.submit(*ops)))
        q.wait(jobs[-1][3])
        for i, crops, out_t, ticket in jobs:
            q.wait(ticket)
            ref = np.full((10, 3 * dst[0] * dst[1]), -777.0, dtype=np.float32)
            s = surfs[i % 3]
            luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, s.ctypes.data, w, owner=s)
            oracle.execute(cvgs.lower(_nv12_ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, capi.YUV_LIMITED, capi.BT709, capi.YUV_NV12)))
            H.assert_bit_exact(out_t.cpu().numpy(), ref, "NV12 batch %d" % i)
        st = q.stats()
        assert st["completed"] == 40 and st["error"] == 0, st
    finally:
        q.destroy()


def test_queue_serves_one_kind(torch_dev):
    """a queue's first submit decides what it serves; the other kind and the layouts the worker does not read are refused"""
    torch, dev = torch_dev
    w, h = 640, 360
    surf_t = torch.from_numpy(H.random_u8((h + h // 2, w), seed=5)).to(dev)
    frame_t = torch.from_numpy(H.random_u8((h, w, 3), seed=6)).to(dev)
    out_t = torch.zeros((1, 3 * 64 * 128), dtype=torch.float32, device=dev)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf_t.data_ptr(), w, owner=surf_t)
    nv = _nv12_ops([luma], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), (64, 128), capi.YUV_FULL, capi.BT709, capi.YUV_NV12)
    px = H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), [(0, 0, w, h)], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), (64, 128), 3)
    for first, second in ((nv, px), (px, nv)):
        q = cvgs.Queue()
        try:
            q.wait(q.submit(*first))
            with pytest.raises(capi.CvgsError):
                q.submit(*second)
            q.wait(q.submit(*first))
            assert q.stats()["error"] == 0
        finally:
            q.destroy()
    q = cvgs.Queue()
    try:
        i420 = torch.from_numpy(H.random_u8((h + h // 2, w), seed=7)).to(dev)
        lu = cvgs.GpuMat(h, w, cvgs.CV_8UC1, i420.data_ptr(), w, owner=i420)
        with pytest.raises(capi.CvgsError):
            q.submit(*_nv12_ops([lu], cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), (64, 128), capi.YUV_FULL, capi.BT709, capi.YUV_I420))
    finally:
        q.destroy()


# ---- the queue's fourth kind: crops of P010 decoder surfaces (10-bit, 16-bit samples: K4's S16 shape) ----------------------------
def _p010_surface(w, h, seed):
    """(H * 3 / 2, W) u16: 10-bit codes in the high bits, the six low bits dirty (a decoder may leave anything there)"""
    return ((H.random_u16((h + h // 2, w), seed) >> 6) << 6 | (H.random_u16((h + h // 2, w), seed + 1) & 63)).astype(np.uint16)


def _p010_ops(luma, crops, out_mat, dst, range_, prim, swap=True, ar=None, background=None, used=None, half=False):
    f, hf = cvgs.CV_32FC3, cvgs.CV_16FC3
    rd = cvgs.read_nv12([luma.nv12_roi(*c) for c in crops], dst, range_, prim, False, layout=capi.YUV_P010)
    if ar is not None:
        rd.ar = ar
    if background is not None:
        rd.background = cvgs._scalar(background)
    if used is not None:
        rd.used_planes = used
    ops = [rd] + ([cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f)] if swap else [])
    ops += [cvgs.multiply(f, [1 / 1023.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]
    return ops + ([cvgs.convertTo(f, hf), cvgs.split(hf, out_mat, dst)] if half else [cvgs.split(f, out_mat, dst)])


def _p010_case(oracle, torch, dev, q, surf, w, h, crops, dst, half=False, **kw):
    """one batch of crops (x, y, w, h; even) of the P010 surface on the queue, on cvgs_execute and on the oracle"""
    n = len(crops)
    surf_t = torch.from_numpy(surf.view(np.int16)).to(dev)
    tt, nt, mt = (torch.float16, np.float16, cvgs.CV_16FC1) if half else (torch.float32, np.float32, cvgs.CV_32FC1)
    outs = {}
    for name in ("queue", "execute"):
        out_t = torch.full((n, 3 * dst[0] * dst[1]), -777.0, dtype=tt, device=dev)
        luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, surf_t.data_ptr(), 2 * w, owner=surf_t)
        ops = _p010_ops(luma, crops, cvgs.GpuMat.from_tensor(out_t, mt), dst, half=half, **kw)
        torch.cuda.synchronize()
        if name == "queue":
            q.wait(q.submit(*ops))
        else:
            cvgs.executeOperations(torch.cuda.current_stream(), *ops)
        torch.cuda.synchronize()
        outs[name] = out_t.cpu().numpy()
    ref = np.full((n, 3 * dst[0] * dst[1]), -777.0, dtype=nt)
    luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, surf.ctypes.data, 2 * w, owner=surf)
    oracle.execute(cvgs.lower(_p010_ops(luma, crops, cvgs.GpuMat.from_array(ref, mt), dst, half=half, **kw)))
    H.assert_bit_exact(outs["queue"], ref, "P010 queue batch vs the oracle %s" % (kw,))
    H.assert_bit_exact(outs["queue"], outs["execute"], "P010 queue batch vs cvgs_execute")


@pytest.mark.parametrize("dst,kw", [
    ((64, 128), dict(range_=capi.YUV_LIMITED, prim=capi.BT2020)),
    ((64, 128), dict(range_=capi.YUV_FULL, prim=capi.BT709, swap=False)),
    ((64, 128), dict(range_=capi.YUV_LIMITED, prim=capi.BT601, half=True)),
    ((100, 37), dict(range_=capi.YUV_FULL, prim=capi.BT2020)),                              # ragged: 2 column tiles, 37 rows
    ((64, 64), dict(range_=capi.YUV_LIMITED, prim=capi.BT2020, ar=cvgs.PRESERVE_AR, background=[114.0, 100.5, 7.25])),
    ((96, 64), dict(range_=capi.YUV_FULL, prim=capi.BT709, ar=cvgs.PRESERVE_AR_LEFT, background=[1.0, 2.0, 3.0], used=4)),
    ((256, 16), dict(range_=capi.YUV_LIMITED, prim=capi.BT2020)),
])
def test_queue_p010_batch_matches_the_oracle(oracle, torch_dev, dst, kw):
    torch, dev = torch_dev
    w, h = 1280, 720
    surf = _p010_surface(w, h, seed=121)
    crops = _even_crops(7, w, h, seed=131) + [(0, 0, w, h), (w - 4, h - 2, 4, 2)]  # the whole surface, a 4 x 2 corner
    q = cvgs.Queue()
    try:
        _p010_case(oracle, torch, dev, q, surf, w, h, crops, dst, **kw)
        assert q.stats()["error"] == 0
    finally:
        q.destroy()


def test_queue_p010_whole_6k_surface(oracle, torch_dev):
    """BASELINE cfg #3's 10-bit sibling through the queue: a 6K P010 surface (BT.2020 limited) -> BGR float -> 1280 x 720 -> normalize
    -> split, frame after frame; then an upscaled crop (taps that share samples and chroma pairs)."""
    torch, dev = torch_dev
    w, h = 6144, 3456
    q = cvgs.Queue()
    try:
        for seed in (141, 142):
            surf = _p010_surface(w, h, seed=seed)
            _p010_case(oracle, torch, dev, q, surf, w, h, [(0, 0, w, h)], (1280, 720), range_=capi.YUV_LIMITED, prim=capi.BT2020)
        _p010_case(oracle, torch, dev, q, surf, w, h, [(6144 - 40, 3456 - 24, 40, 24), (0, 0, 6, 4)], (200, 120), range_=capi.YUV_LIMITED, prim=capi.BT2020)
        assert q.stats()["error"] == 0
    finally:
        q.destroy()


def test_queue_p010_many_batches_in_flight_and_kind_latch(oracle, torch_dev):
    """40 P010 batches submitted back to back (every task size), each checked; a P010 queue refuses NV12 batches and vice versa"""
    torch, dev = torch_dev
    w, h, dst, n = 1920, 1080, (64, 128), 12
    surf = _p010_surface(w, h, seed=151)
    surf_t = torch.from_numpy(surf.view(np.int16)).to(dev)
    luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, surf_t.data_ptr(), 2 * w, owner=surf_t)
    hl = cvgs.GpuMat(h, w, cvgs.CV_16UC1, surf.ctypes.data, 2 * w, owner=surf)
    q = cvgs.Queue()
    try:
        outs, lists, tickets = [], [], []
        torch.cuda.synchronize()
        for i in range(40):
            crops = _even_crops(n, w, h, seed=200 + i)
            out_t = torch.zeros((n, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
            tickets.append(q.submit(*_p010_ops(luma, crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), dst, capi.YUV_LIMITED, capi.BT2020)))
            outs.append(out_t)
            lists.append(crops)
        q.wait(tickets[-1])
        torch.cuda.synchronize()
        for i in range(40):
            ref = np.zeros((n, 3 * dst[0] * dst[1]), dtype=np.float32)
            oracle.execute(cvgs.lower(_p010_ops(hl, lists[i], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, capi.YUV_LIMITED, capi.BT2020)))
            H.assert_bit_exact(outs[i].cpu().numpy(), ref, "P010 batch %d" % i)
        nv = torch.from_numpy(H.random_u8((h + h // 2, w), seed=9)).to(dev)
        l8 = cvgs.GpuMat(h, w, cvgs.CV_8UC1, nv.data_ptr(), w, owner=nv)
        with pytest.raises(capi.CvgsError):
            q.submit(*_nv12_ops([l8], cvgs.GpuMat.from_tensor(outs[0], cvgs.CV_32FC1), dst, capi.YUV_FULL, capi.BT709, capi.YUV_NV12))
        assert q.stats()["error"] == 0
    finally:
        q.destroy()
    q = cvgs.Queue()
    try:
        q.wait(q.submit(*_nv12_ops([l8], cvgs.GpuMat.from_tensor(outs[0], cvgs.CV_32FC1), dst, capi.YUV_FULL, capi.BT709, capi.YUV_NV12)))
        with pytest.raises(capi.CvgsError):
            q.submit(*_p010_ops(luma, lists[0], cvgs.GpuMat.from_tensor(outs[1], cvgs.CV_32FC1), dst, capi.YUV_LIMITED, capi.BT2020))
    finally:
        q.destroy()


def test_queue_two_host_threads_and_destroy_with_batches_in_flight(oracle, torch_dev):
    """submits are serialised by the queue's mutex: two host threads feeding ONE queue get every batch right; destroying a queue
    with batches in flight completes them first (the workers drain the ring before they see the stop word)"""
    import threading
    torch, dev = torch_dev
    frame = H.random_u8((720, 1280, 3), seed=77)
    frame_t = torch.from_numpy(frame).to(dev)
    q = cvgs.Queue(depth=16)
    jobs, errors = [], []
    lock = threading.Lock()

    def feed(tid):
        try:
            for i in range(30):
                crops = H.random_crops(8, 1280, 720, wmax=300, hmax=400, seed=1000 * tid + i)
                out_t, ops = gpu_chain(torch, dev, frame_t, crops, 8, (64, 128), 3)
                ticket = q.submit(*ops)
                with lock:
                    jobs.append((crops, out_t, ticket))
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    torch.cuda.synchronize()
    threads = [threading.Thread(target=feed, args=(t,)) for t in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(jobs) == 60 and sorted(j[2] for j in jobs) == list(range(60))
    q.destroy()  # no wait in front of it: up to 16 batches are still in flight
    torch.cuda.synchronize()
    for crops, out_t, ticket in jobs:
        H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, 8, (64, 128), 3), "ticket %d" % ticket)


@pytest.mark.parametrize("cn,dst,kw", [(3, (64, 128), {}), (4, (64, 128), {}), (3, (100, 37), {"ar": cvgs.PRESERVE_AR, "background": [9.0, 8.0, 7.0, 0.0]}),
                                       (3, (64, 128), {"used": 5, "background": [7.0, 8.0, 9.0, 0.0]})])
def test_queue_fp16_tensor_matches_the_oracle(oracle, torch_dev, cn, dst, kw):
    """the half-precision hand-off through the queue: the chain ends with convertTo<CV_32FCn, CV_16FCn>, the worker's stores convert"""
    torch, dev = torch_dev
    frame = H.random_u8((1080, 1920, cn), seed=13)
    crops = H.random_crops(9, 1920, 1080, wmax=400, hmax=500, seed=17)
    frame_t = torch.from_numpy(frame).to(dev)
    q = cvgs.Queue()
    try:
        out_t = torch.full((9, cn * dst[0] * dst[1]), -7.0, dtype=torch.float16, device=dev)
        ops = H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.make_type(cvgs.CV_8U, cn)), crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_16FC1), dst, cn, half=True, **kw)
        torch.cuda.synchronize()
        q.wait(q.submit(*ops))
        torch.cuda.synchronize()
        ref = np.full((9, cn * dst[0] * dst[1]), -7.0, dtype=np.float16)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_16FC1), dst, cn, half=True, **kw)))
        H.assert_bit_exact(out_t.cpu().numpy(), ref, "fp16 queue batch %s %s" % (dst, kw))
    finally:
        q.destroy()


def test_queue_nv12_fp16_tensor(oracle, torch_dev):
    torch, dev = torch_dev
    w, h, dst = 1280, 720, (96, 54)
    surf = H.random_u8((h + h // 2, w), seed=23)
    crops = _even_crops(6, w, h, seed=33)
    f, hf = cvgs.CV_32FC3, cvgs.CV_16FC3

    def ops_for(luma, out_mat):
        return [cvgs.read_nv12([luma.nv12_roi(*c) for c in crops], dst, capi.YUV_LIMITED, capi.BT709, False), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                cvgs.convertTo(f, hf), cvgs.split(hf, out_mat, dst)]

    st = torch.from_numpy(surf).to(dev)
    out_t = torch.zeros((6, 3 * dst[0] * dst[1]), dtype=torch.float16, device=dev)
    q = cvgs.Queue()
    try:
        torch.cuda.synchronize()
        q.wait(q.submit(*ops_for(cvgs.GpuMat(h, w, cvgs.CV_8UC1, st.data_ptr(), w, owner=st), cvgs.GpuMat.from_tensor(out_t, cvgs.CV_16FC1))))
        torch.cuda.synchronize()
        ref = np.zeros((6, 3 * dst[0] * dst[1]), dtype=np.float16)
        oracle.execute(cvgs.lower(ops_for(cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf.ctypes.data, w, owner=surf), cvgs.GpuMat.from_array(ref, cvgs.CV_16FC1))))
        H.assert_bit_exact(out_t.cpu().numpy(), ref, "NV12 fp16 queue batch")
    finally:
        q.destroy()


def test_queue_two_queues_never_wait_for_each_others_slots(oracle, torch_dev):
    """A server's workgroups must all be resident, two servers do not fit the chip together, and two PARTLY resident servers can
    wait for each other forever (a retirement and a relaunch racing for the freed slots: the watchdog's 250 ms, once in ~10 runs of
    the test above).  One server per device at a time: a queue that needs to launch asks the current server to retire first.
    Provoked here: short idle times, sleeps around them, a pixel queue and an NV12 queue alternating, then two threads."""
    import threading
    import time
    torch, dev = torch_dev
    frame = H.random_u8((720, 1280, 3), seed=61)
    frame_t = torch.from_numpy(frame).to(dev)
    w, h = 1280, 720
    surf = H.random_u8((h + h // 2, w), seed=62)
    surf_t = torch.from_numpy(surf).to(dev)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf_t.data_ptr(), w, owner=surf_t)
    crops = H.random_crops(12, 1280, 720, wmax=300, hmax=400, seed=63)
    ecrops = _even_crops(6, w, h, seed=64)
    out_a, ops_a = gpu_chain(torch, dev, frame_t, crops, 12, (64, 128), 3)
    out_b = torch.zeros((6, 3 * 64 * 128), dtype=torch.float32, device=dev)
    ops_b = _nv12_ops([luma.nv12_roi(*c) for c in ecrops], cvgs.GpuMat.from_tensor(out_b, cvgs.CV_32FC1), (64, 128), capi.YUV_FULL, capi.BT709, capi.YUV_NV12)
    la, lb = cvgs.lower(ops_a), cvgs.lower(ops_b)
    ref_a = oracle_out(oracle, frame, crops, 12, (64, 128), 3)
    ref_b = np.zeros((6, 3 * 64 * 128), dtype=np.float32)
    hl = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf.ctypes.data, w, owner=surf)
    oracle.execute(cvgs.lower(_nv12_ops([hl.nv12_roi(*c) for c in ecrops], cvgs.GpuMat.from_array(ref_b, cvgs.CV_32FC1), (64, 128), capi.YUV_FULL, capi.BT709, capi.YUV_NV12)))
    qa, qb = cvgs.Queue(idle_us=30.0), cvgs.Queue(idle_us=30.0)
    try:
        torch.cuda.synchronize()
        rng = np.random.default_rng(5)
        ta = tb = None
        for i in range(400):  # one thread, alternating, with pauses around the idle time: retirements and relaunches interleave
            ta = qa.submit_lowered(la)
            if i % 3 == 0:
                time.sleep(float(rng.uniform(0, 120e-6)))
            tb = qb.submit_lowered(lb)
            if i % 5 == 0:
                time.sleep(float(rng.uniform(0, 120e-6)))
        qa.wait(ta)
        qb.wait(tb)
        errors = []

        def feed(q, lowered, n):
            try:
                t = None
                for i in range(n):
                    t = q.submit_lowered(lowered)
                    if i % 16 == 15:
                        q.wait(t)
                q.wait(t)
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)

        threads = [threading.Thread(target=feed, args=(qa, la, 600)), threading.Thread(target=feed, args=(qb, lb, 600))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert qa.stats()["error"] == 0 and qb.stats()["error"] == 0
        torch.cuda.synchronize()
        H.assert_bit_exact(out_a.cpu().numpy(), ref_a, "pixel queue")
        H.assert_bit_exact(out_b.cpu().numpy(), ref_b, "NV12 queue")
    finally:
        qa.destroy()
        qb.destroy()


@pytest.mark.parametrize("depth,cn,dst,kw", [("16U", 3, (64, 128), {}), ("16S", 4, (64, 128), {}), ("16U", 4, (100, 37), {"ar": cvgs.PRESERVE_AR, "background": [9.0, 8.0, 7.0, 6.0]}),
                                             ("16S", 3, (64, 128), {"used": 5, "background": [7.0, 8.0, 9.0, 0.0], "swap": False})])
def test_queue_16_bit_pixel_crops_match_the_oracle(oracle, torch_dev, depth, cn, dst, kw):
    """the queue's third kind: CV_16U / CV_16S C3 / C4 crops -- the other source types of the reference's K1 sweep
    (tests/batchresize/test_batchresize_x_split3D.cu:427-432); a queue of its own (the first submit decides)"""
    torch, dev = torch_dev
    sd = cvgs.CV_16U if depth == "16U" else cvgs.CV_16S
    frame = H.random_u16((720, 1280, cn), seed=29).view(np.uint16 if depth == "16U" else np.int16)
    crops = H.random_crops(9, 1280, 720, wmax=400, hmax=500, seed=31) + [(1276, 0, 4, 700)]  # + a crop barely wider than one tap window
    n = len(crops)
    frame_t = torch.from_numpy(frame.view(np.int16)).to(dev)
    q = cvgs.Queue()
    try:
        out_t = torch.full((n, cn * dst[0] * dst[1]), -777.0, dtype=torch.float32, device=dev)
        ops = H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.make_type(sd, cn)), crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), dst, cn, src_depth=sd, **kw)
        torch.cuda.synchronize()
        for _ in range(3):
            t = q.submit(*ops)
        q.wait(t)
        torch.cuda.synchronize()
        ref = np.full((n, cn * dst[0] * dst[1]), -777.0, dtype=np.float32)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(sd, cn)), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, cn, src_depth=sd, **kw)))
        H.assert_bit_exact(out_t.cpu().numpy(), ref, "16-bit queue batch %s C%d %s %s" % (depth, cn, dst, kw))
        # the other kinds are refused on this queue
        u8 = torch.zeros((64, 64, 3), dtype=torch.uint8, device=dev)
        o8 = torch.zeros((1, 3 * 64 * 128), dtype=torch.float32, device=dev)
        with pytest.raises(capi.CvgsError):
            q.submit(*H.k1_chain(cvgs.GpuMat.from_tensor(u8, cvgs.CV_8UC3), [(0, 0, 64, 64)], cvgs.GpuMat.from_tensor(o8, cvgs.CV_32FC1), (64, 128), 3))
        assert q.stats()["error"] == 0
    finally:
        q.destroy()


def test_queue_wait_on_a_ticket_covers_every_earlier_batch(oracle, torch_dev):
    """Batches finish in any order on the device (the last of 24 60-crop batches are still running when the one-crop batches submitted
    behind them are done): a wait for the LAST ticket must still mean "everything up to here is complete" -- on the host (the completed count is read
    right after the wait, no device synchronisation) and on a consumer stream (it copies the BIG batch's tensor)."""
    torch, dev = torch_dev
    q = cvgs.Queue()
    try:
        frame = H.random_u8((1080, 1920, 3), seed=61)
        frame_t = torch.from_numpy(frame).to(dev)
        big_crops = H.random_crops(60, 1920, 1080, wmax=500, hmax=600, seed=62)
        big_t, big_ops = gpu_chain(torch, dev, frame_t, big_crops, 60, (64, 128), 3)
        big = cvgs.lower(big_ops)
        small = []
        for i in range(8):
            crops = H.random_crops(1, 1920, 1080, wmax=64, hmax=64, seed=70 + i)
            out_t, ops = gpu_chain(torch, dev, frame_t, crops, 1, (64, 16), 3)
            small.append((crops, out_t, cvgs.lower(ops)))
        want_big = oracle_out(oracle, frame, big_crops, 60, (64, 128), 3)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        submitted = 0
        for rep in range(40):
            big_t.fill_(-5.0)
            torch.cuda.synchronize()
            for _ in range(24):
                q.submit_lowered(big)
            last = None
            for _, _, lw in small:
                last = q.submit_lowered(lw)
            submitted += 32
            if rep % 2 == 0:
                q.wait(last)
                st = q.stats()
                assert st["completed"] == submitted and st["error"] == 0, (rep, st)
                got = big_t.cpu().numpy()
            else:
                q.stream_wait(last, s)
                with torch.cuda.stream(s):
                    copy = big_t.clone()
                s.synchronize()
                got = copy.cpu().numpy()
            H.assert_bit_exact(got, want_big, "the big batch behind the last ticket, rep %d" % rep)
        q.wait(last)
        for crops, out_t, _ in small:
            H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, 1, (64, 16), 3), "one-crop batch")
    finally:
        q.destroy()


@pytest.mark.parametrize("n,cn,kw", [
    (150, 3, {}),                                                     # three ring slots: 74 + 74 + 2
    (75, 4, {}),                                                      # a second slot with ONE plane
    (200, 3, {"used": 120, "background": [7.0, 8.0, 9.0, 0.0]}),      # the used / default-value boundary falls inside the second slot
    (148, 3, {"ar": cvgs.PRESERVE_AR, "background": [114.0, 114.0, 114.0, 0.0], "swap": False}),
    (300, 3, {"half": True}),                                         # the reference's largest batch (test_batchresize_x_split3D.cu:384-392), fp16 tensor
])
def test_queue_batches_larger_than_a_ring_slot(oracle, torch_dev, n, cn, kw):
    """A ring slot holds 74 planes; larger batches go out as consecutive slots over slices of the tensor, one ticket."""
    torch, dev = torch_dev
    frame = H.random_u8((1080, 1920, cn), seed=90 + n)
    crops = H.random_crops(n, 1920, 1080, wmax=300, hmax=300, seed=91 + n)
    half = kw.get("half", False)
    dst = (64, 32)
    q = cvgs.Queue()
    try:
        ft = torch.from_numpy(frame).to(dev)
        out_t = torch.full((n, cn * dst[0] * dst[1]), -3.0, dtype=torch.float16 if half else torch.float32, device=dev)
        ref = np.full((n, cn * dst[0] * dst[1]), -3.0, dtype=np.float16 if half else np.float32)
        t_out = cvgs.CV_16FC1 if half else cvgs.CV_32FC1
        ops = H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.make_type(cvgs.CV_8U, cn)), crops, cvgs.GpuMat.from_tensor(out_t, t_out), dst, cn, **kw)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), crops, cvgs.GpuMat.from_array(ref, t_out), dst, cn, **kw)))
        torch.cuda.synchronize()
        before = q.stats()["submitted"]
        q.wait(q.submit(*ops))
        st = q.stats()
        assert st["submitted"] - before == (n + 73) // 74 and st["completed"] == st["submitted"], st
        H.assert_bit_exact(out_t.cpu().numpy().view(np.uint16 if half else np.uint32), ref.view(np.uint16 if half else np.uint32), "%d crops over ring slots" % n)
    finally:
        q.destroy()
