"""cvgs_execute_many as a TICK -- the frames of several cameras, each with its crop list and its tensor, in ONE fused launch -- over the
shapes a serving loop meets: every segment-table size up to CVGS_MAX_CHAINS, 3 / 4 channels, with and without the R <-> B swap, fp16
tensors, aspect-ratio windows and default planes, ragged targets, sources narrower than a tap window, CNHW tensors; bit-exact against
the CPU oracle.  Reference call shape: one executeOperations per frame, include/cvGPUSpeedup.cuh:464-473; a tick = the calls of several
cameras (tests/batchresize/test_batchresize_x_split3D.cu:384-392 sweeps the batch the same way).  Also: the table slots' recycling
through the fused launch's progress word (hundreds of ticks with a different crop list each, nothing synchronised in between; round 5:
no HIP event behind the launch), two streams at once, captured launches with device tables, and -- in a subprocess, the knob is read
once -- the same ticks with the event-tracked descriptor scratch (CVGS_MANY_PROGRESS_WORD=0): one digest."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chain(torch, dev, frame_t, crops, dst, cn, half=False, table=False, out=None, **kw):
    n = len(crops)
    if out is None:
        out = torch.full((n, cn * dst[0] * dst[1]), -777.0, dtype=torch.float16 if half else torch.float32, device=dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, cvgs.make_type(cvgs.CV_8U, cn))
    g_out = cvgs.GpuMat.from_tensor(out, cvgs.CV_16FC1 if half else cvgs.CV_32FC1)
    ops = H.k1_chain(g_src, crops, g_out, dst, cn, half=half, **kw)
    keep = None
    if table:
        keep = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(dev)
        ops = H.k1_chain(g_src, crops, g_out, dst, cn, half=half, table=keep.data_ptr(), **kw)
    return ops, out, keep


def _oracle(oracle, frame, crops, dst, cn, half=False, **kw):
    ref = np.full((len(crops), cn * dst[0] * dst[1]), -777.0, dtype=np.float16 if half else np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), crops,
                                         cvgs.GpuMat.from_array(ref, cvgs.CV_16FC1 if half else cvgs.CV_32FC1), dst, cn, half=half, **kw)))
    return ref


def _tick(torch, dev, oracle, lib, n_chains, cn=3, dst=(64, 128), half=False, table=False, seed=1, frame_hw=(540, 960), crops_lo=1, crops_hi=40,
          stream=None, **kw):
    rng = np.random.default_rng(seed)
    chains, outs, keeps, refs = [], [], [], []
    fh, fw = frame_hw
    for m in range(n_chains):
        frame = H.random_u8((fh, fw, cn), seed=seed * 1000 + m)
        n = int(rng.integers(crops_lo, crops_hi + 1))
        crops = H.random_crops(n, fw, fh, seed=seed * 77 + m, wmax=min(400, fw), hmax=min(500, fh))
        ft = torch.from_numpy(frame).to(dev)
        ops, out, keep = _chain(torch, dev, ft, crops, dst, cn, half=half, table=table, **kw)
        chains.append(ops)
        outs.append(out)
        keeps += [ft, keep]
        refs.append(_oracle(oracle, frame, crops, dst, cn, half=half, **kw))
    lowered, arr = cvgs.executeMany(stream or torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    took = 1
    for m in range(n_chains):
        H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "tick of %d chains, chain %d" % (n_chains, m))
    return took, chains, outs, refs, keeps


@pytest.mark.parametrize("n_chains", [2, 3, 16, 17, 64, 65, 128])
def test_tick_matches_the_oracle_for_every_segment_table_size(oracle, device, lib, n_chains):
    import torch
    took, *_ = _tick(torch, device, oracle, lib, n_chains, seed=10 + n_chains, crops_hi=12 if n_chains > 16 else 40)
    assert took == 1


@pytest.mark.parametrize("cn,swap,half", [(3, True, False), (3, False, False), (4, True, False), (4, False, True), (3, True, True)])
def test_tick_channels_programs_and_fp16(oracle, device, lib, cn, swap, half):
    import torch
    took, *_ = _tick(torch, device, oracle, lib, 6, cn=cn, half=half, swap=swap, seed=40 + cn + 2 * swap + 4 * half)
    assert took == 1


@pytest.mark.parametrize("kw", [{"ar": cvgs.PRESERVE_AR, "background": [128.0, 64.0, 32.0, 0.0]},
                                {"ar": cvgs.PRESERVE_AR_RN_EVEN, "background": [1.0, 2.0, 3.0, 0.0]},
                                {"used": 2, "background": [7.0, 8.0, 9.0, 0.0]}])
def test_tick_aspect_ratio_windows_and_default_planes(oracle, device, lib, kw):
    import torch
    took, *_ = _tick(torch, device, oracle, lib, 5, seed=70, crops_lo=3, crops_hi=9, **kw)
    assert took == 1


@pytest.mark.parametrize("dst", [(100, 37), (64, 4), (1, 1), (200, 130), (63, 129)])
def test_tick_ragged_targets(oracle, device, lib, dst):
    """targets that are not multiples of the 4 x 64 tile: ragged column tiles, row groups that end inside a task"""
    import torch
    took, *_ = _tick(torch, device, oracle, lib, 4, dst=dst, seed=90 + dst[0], crops_hi=10)
    assert took == 1


@pytest.mark.parametrize("table", [False, True])
def test_tick_sources_narrower_than_a_tap_window(oracle, device, lib, table):
    """crops 1-2 pixels wide / 1 row high: the byte-by-byte gather (the server refuses such crops on the host; a device table cannot be looked at)"""
    import torch
    fh, fw = 64, 96
    chains, outs, refs, keep = [], [], [], []
    for m in range(3):
        frame = H.random_u8((fh, fw, 3), seed=500 + m)
        crops = [(5, 3, 1, 1), (7, 9, 2, 30), (0, 0, 1, 64), (94, 10, 2, 2), (10, 10, 3, 1), (20, 5, 40, 50)]
        ft = torch.from_numpy(frame).to(device)
        ops, out, kp = _chain(torch, device, ft, crops, (64, 128), 3, table=table)
        chains.append(ops)
        outs.append(out)
        keep += [ft, kp]
        refs.append(_oracle(oracle, frame, crops, (64, 128), 3))
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for m in range(3):
        H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "tiny sources, chain %d" % m)


def test_tick_cnhw_tensor(oracle, device, lib):
    """TensorTSplit (CNHW): the channel stride is the tensor's N"""
    import torch
    fh, fw = 300, 400
    chains, outs, refs, keep = [], [], [], []
    for m in range(3):
        frame = H.random_u8((fh, fw, 3), seed=600 + m)
        crops = H.random_crops(6, fw, fh, seed=610 + m, wmax=200, hmax=200)
        n = len(crops)
        ft = torch.from_numpy(frame).to(device)
        out = torch.full((3, n, 128 * 64), -777.0, dtype=torch.float32, device=device)
        ref = np.full((3, n, 128 * 64), -777.0, dtype=np.float32)
        f = cvgs.CV_32FC3

        def ops_for(src_mat, data_ptr, owner):
            return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [src_mat.roi(*c) for c in crops], (64, 128), n),
                    cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, [1.0, 4.0, 3.2]), cvgs.divide(f, [3.2, 0.6, 11.8]),
                    cvgs.splitT(f, data_ptr, 64, 128, n, keep=owner)]
        chains.append(ops_for(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), out.data_ptr(), out))
        oracle.execute(cvgs.lower(ops_for(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), ref.ctypes.data, ref)))
        outs.append(out)
        refs.append(ref)
        keep.append(ft)
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for m in range(3):
        H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "CNHW tick, chain %d" % m)


def test_hundreds_of_ticks_recycle_their_table_slots(oracle, device, lib):
    """300 ticks back to back on one stream, a DIFFERENT crop list in every tick, no synchronisation in between: the pinned table slots
    are recycled by the launches' completion word while the host runs ahead; every tick's tensors are checked at the end (a slot
    rewritten while its kernel still reads it would show up as another tick's crops)."""
    import torch
    fh, fw = 360, 640
    frames = [H.random_u8((fh, fw, 3), seed=700 + k) for k in range(3)]
    fts = [torch.from_numpy(f).to(device) for f in frames]
    ticks, n_ticks = [], 300
    s = torch.cuda.Stream()
    for i in range(n_ticks):
        chains, outs, meta = [], [], []
        for k in range(3):
            crops = H.random_crops(4 + (i + k) % 5, fw, fh, seed=7000 + 10 * i + k, wmax=300, hmax=300)
            ops, out, _ = _chain(torch, device, fts[k], crops, (64, 128), 3)
            chains.append(ops)
            outs.append(out)
            meta.append((k, crops))
        torch.cuda.current_stream().synchronize()  # the tensors' fills ran on torch's stream: they must have landed before a tick on `s` writes them (`s` itself is never waited for)
        ticks.append((outs, meta, cvgs.executeMany(s, chains)))
    s.synchronize()
    for i in (list(range(0, n_ticks, 7)) + [n_ticks - 1]):
        outs, meta, _ = ticks[i]
        for out, (k, crops) in zip(outs, meta):
            H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frames[k], crops, (64, 128), 3), "tick %d, camera %d" % (i, k))


def test_two_streams_tick_at_once(oracle, device, lib):
    """every stream has its own counter block: ticks of two streams overlap on the device"""
    import torch
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    jobs = []
    for i in range(40):
        st = s1 if i % 2 == 0 else s2
        rng_seed = 800 + i
        fh, fw = 540, 960
        chains, outs, refs = [], [], []
        for m in range(4):
            frame = H.random_u8((fh, fw, 3), seed=rng_seed * 10 + m)
            crops = H.random_crops(20, fw, fh, seed=rng_seed * 13 + m, wmax=400, hmax=500)
            ft = torch.from_numpy(frame).to(device)
            ops, out, _ = _chain(torch, device, ft, crops, (64, 128), 3)
            chains.append(ops)
            outs.append(out)
            refs.append((frame, crops, ft))
        jobs.append((st, chains, outs, refs))
    torch.cuda.synchronize()
    held = [cvgs.executeMany(st, chains) for st, chains, _, _ in jobs]
    s1.synchronize()
    s2.synchronize()
    for i in (0, 1, 17, 38, 39):
        _, _, outs, refs = jobs[i]
        for out, (frame, crops, _) in zip(outs, refs):
            H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frame, crops, (64, 128), 3), "two streams, job %d" % i)
    assert held


def test_captured_tick_with_device_tables_replays(oracle, device, lib):
    """device plane tables: the tick launch is capturable (a counter block of its own per captured launch) and replays"""
    import torch
    took, chains, outs, refs, keeps = _tick(torch, device, oracle, lib, 8, table=True, seed=900)
    assert took == 1
    lowered = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(lowered)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(3):  # three fused launches in one graph
                capi.check(lib.cvgs_execute_many(arr, len(lowered), torch.cuda.current_stream().cuda_stream))
        for rep in range(3):
            for o in outs:
                o.fill_(-5.0)
            g.replay()
            torch.cuda.synchronize()
            for m in range(len(outs)):
                H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "captured tick, replay %d, chain %d" % (rep, m))


def test_interpreted_programs_in_a_tick(oracle, device, lib):
    """a program that is not [swap] mul sub div: the fused launch's interpreted instantiation, same bits"""
    import torch
    fh, fw = 300, 400
    chains, outs, refs, keep = [], [], [], []
    f = cvgs.CV_32FC3
    for m in range(3):
        frame = H.random_u8((fh, fw, 3), seed=950 + m)
        crops = H.random_crops(6, fw, fh, seed=960 + m, wmax=200, hmax=200)
        ft = torch.from_numpy(frame).to(device)
        out = torch.full((6, 3 * 128 * 64), -777.0, dtype=torch.float32, device=device)
        ref = np.full((6, 3 * 128 * 64), -777.0, dtype=np.float32)

        def ops_for(src_mat, out_mat):  # an ADD stage: not [swap] mul sub div
            return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [src_mat.roi(*c) for c in crops], (64, 128), 6),
                    cvgs.multiply(f, [0.5] * 3), cvgs.add(f, [1.0, 2.0, 3.0]), cvgs.split(f, out_mat, (64, 128))]
        chains.append(ops_for(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)))
        oracle.execute(cvgs.lower(ops_for(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
        outs.append(out)
        refs.append(ref)
        keep.append(ft)
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for m in range(3):
        H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "interpreted program, chain %d" % m)


_GRID_VS_TICK = r"""
import os, sys, hashlib
sys.path.insert(0, %(root)r)
import numpy as np, torch
from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
dev = torch.device("cuda:0")
lib = capi.load_library()
h = hashlib.sha256()
for seed in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
    chains, outs, keep = [], [], []
    for m in range(9):
        frame = H.random_u8((540, 960, 3), seed=seed * 100 + m)
        crops = H.random_crops(3 + 4 * m, 960, 540, seed=seed * 31 + m, wmax=400, hmax=500)
        ft = torch.from_numpy(frame).to(dev)
        out = torch.full((len(crops), 3 * 64 * 128), -777.0, dtype=torch.float32, device=dev)
        chains.append(H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)))
        outs.append(out); keep.append(ft)
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for o in outs:
        h.update(o.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def test_inline_ring_and_event_tracked_tables_write_the_same_bytes():
    """the same ticks in three fresh processes: descriptors inside the kernel arguments (the default for <= 1024 planes), in the stream's
    table ring recycled through the launch's progress word (CVGS_MANY_INLINE=0), and in the event-tracked descriptor scratch
    (+ CVGS_MANY_PROGRESS_WORD=0): one digest"""
    res = {}
    for mode, extra in (("inline", {}), ("ring", {"CVGS_MANY_INLINE": "0"}), ("events", {"CVGS_MANY_INLINE": "0", "CVGS_MANY_PROGRESS_WORD": "0"})):
        env = dict(os.environ)
        env.update(extra)
        p = subprocess.run([sys.executable, "-c", _GRID_VS_TICK % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("DIGEST")]
        assert line, p.stderr[-2000:]
        res[mode] = line[-1].split()[1]
    assert res["inline"] == res["ring"] == res["events"]


def test_the_table_ring_still_serves_this_file():
    """Ticks of up to 1024 host-described planes travel in the kernel arguments since round 5; larger ones, K4 ticks and 1-2 channel sources
    keep the per-stream table ring.  The tests of this file that were written for the ring (slot recycling over hundreds of ticks, two
    streams, four host threads, streams that come and go or are destroyed with work pending) run again with CVGS_MANY_INLINE=0."""
    env = dict(os.environ)
    env["CVGS_MANY_INLINE"] = "0"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "recycle or two_streams or host_threads or come_and_go or destroyed or segment_table_size"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]


@pytest.mark.parametrize("planes_per_chain,n_chains", [(16, 16), (16, 17), (64, 16), (60, 17), (64, 17)])
def test_ticks_around_the_inline_block_sizes(oracle, device, lib, planes_per_chain, n_chains):
    """256 / 272 / 1024 / 1020 / 1088 planes in all: the small block's last fit, the large block's first, its last fits, the ring's first"""
    import torch
    took, *_ = _tick(torch, device, oracle, lib, n_chains, seed=300 + n_chains + planes_per_chain, crops_lo=planes_per_chain, crops_hi=planes_per_chain,
                     frame_hw=(360, 640), dst=(32, 48))
    assert took == 1


def test_a_host_described_tick_is_capturable(oracle, device, lib):
    """descriptors in the kernel arguments: the fused launch of host-described chains is captured into a graph as it is (round 4: refused)
    and replays; a tick beyond 1024 planes under capture runs chain by chain (<= 64 planes each in kernel arguments) -- same bits"""
    import torch
    for n_chains, per in ((6, 20), (16, 50), (20, 60)):  # 120 / 800 planes: one fused launch (16 KB / 52 KB argument block); 1200 planes: chain by chain under capture
        fh, fw = 360, 640
        chains, outs, refs, keep = [], [], [], []
        for m in range(n_chains):
            frame = H.random_u8((fh, fw, 3), seed=1500 + m)
            crops = H.random_crops(per, fw, fh, seed=1510 + m, wmax=300, hmax=300)
            ft = torch.from_numpy(frame).to(device)
            ops, out, _ = _chain(torch, device, ft, crops, (32, 48), 3)
            chains.append(ops)
            outs.append(out)
            keep.append(ft)
            refs.append(_oracle(oracle, frame, crops, (32, 48), 3))
        lowered = [cvgs.lower(ops) for ops in chains]
        arr = cvgs.pack_chains(lowered)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                capi.check(lib.cvgs_execute_many(arr, len(lowered), torch.cuda.current_stream().cuda_stream))
            for rep in range(2):
                for o in outs:
                    o.fill_(-5.0)
                g.replay()
                torch.cuda.synchronize()
                for m in range(n_chains):
                    H.assert_bit_exact(outs[m].cpu().numpy(), refs[m], "captured host-described tick of %d chains, replay %d, chain %d" % (n_chains, rep, m))


def test_ticks_on_streams_that_come_and_go(oracle, device, lib):
    """The per-stream table rings of the fused launch are keyed by the stream HANDLE and never touch a stream after the call that used it:
    400 raw HIP streams are created, run three ticks each (a different crop list per tick) and are destroyed -- the runtime hands handles
    out again, the ring bookkeeping must neither fault nor mix tables up (round 4's event-based reclaim faulted on destroyed streams)."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    fh, fw = 270, 480
    frames = [H.random_u8((fh, fw, 3), seed=1200 + k) for k in range(2)]
    fts = [torch.from_numpy(f).to(device) for f in frames]
    torch.cuda.synchronize()
    checked = 0
    for i in range(400):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        held = []
        for t in range(3):
            chains, outs, meta = [], [], []
            for k in range(2):
                crops = H.random_crops(3 + (i + t + k) % 4, fw, fh, seed=12000 + 31 * i + 7 * t + k, wmax=200, hmax=200)
                ops, out, _ = _chain(torch, device, fts[k], crops, (64, 128), 3)
                chains.append(ops)
                outs.append(out)
                meta.append((k, crops))
            lowered = [cvgs.lower(ops) for ops in chains]
            arr = cvgs.pack_chains(lowered)
            torch.cuda.current_stream().synchronize()  # (the tensors' fills, on torch's stream, before the tick on the raw stream)
            capi.check(lib.cvgs_execute_many(arr, len(lowered), s))
            held.append((outs, meta, lowered, arr))
        assert hip.hipStreamSynchronize(s) == 0
        if i % 40 == 0:
            for outs, meta, _, _ in held:
                for out, (k, crops) in zip(outs, meta):
                    H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frames[k], crops, (64, 128), 3), "stream %d" % i)
                    checked += 1
        assert hip.hipStreamDestroy(s) == 0
    assert checked == 60


def test_ticks_from_several_host_threads(oracle, device, lib):
    """Four host threads tick at once: two of them share ONE stream (the stream's ring is handed out under its mutex: sequence numbers reach
    the stream in order), the other two own a stream each; a different crop list per tick, nothing synchronised until the end."""
    import threading

    import torch
    fh, fw = 360, 640
    frames = [H.random_u8((fh, fw, 3), seed=1300 + k) for k in range(2)]
    fts = [torch.from_numpy(f).to(device) for f in frames]
    shared, own_a, own_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    plan = [(shared, 0), (shared, 1), (own_a, 2), (own_b, 3)]
    n_ticks = 120
    jobs = []  # per thread: [(chains, outs, meta)]
    for _, tid in plan:
        mine = []
        for i in range(n_ticks):
            chains, outs, meta = [], [], []
            for k in range(2):
                crops = H.random_crops(3 + (i + k + tid) % 6, fw, fh, seed=13000 + 1000 * tid + 10 * i + k, wmax=300, hmax=300)
                ops, out, _ = _chain(torch, device, fts[k], crops, (64, 128), 3)
                chains.append(ops)
                outs.append(out)
                meta.append((k, crops))
            lowered = [cvgs.lower(ops) for ops in chains]
            mine.append((lowered, cvgs.pack_chains(lowered), outs, meta))
        jobs.append(mine)
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(len(plan))

    def worker(stream, tid):
        try:
            start.wait()
            for lowered, arr, _, _ in jobs[tid]:
                capi.check(lib.cvgs_execute_many(arr, len(lowered), stream.cuda_stream))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=p) for p in plan]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for tid in range(len(plan)):
        for i in (0, 1, 2, 9, 10, 11, 57, 118, 119):
            _, _, outs, meta = jobs[tid][i]
            for out, (k, crops) in zip(outs, meta):
                H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frames[k], crops, (64, 128), 3), "thread %d, tick %d" % (tid, i))


def test_a_stream_destroyed_with_its_tick_still_pending(oracle, device, lib):
    """hipStreamDestroy right behind a tick that has not started yet (a 3 ms occupancy kernel in front of it), then a NEW stream -- the runtime
    may hand the handle out again -- ticks three times at once.  The ring is keyed by the handle: the old stream's table must not be
    rewritten before its kernel has read it (every tick of every stream is checked)."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    fh, fw = 270, 480
    frames = [H.random_u8((fh, fw, 3), seed=1400 + k) for k in range(2)]
    fts = [torch.from_numpy(f).to(device) for f in frames]
    torch.cuda.synchronize()
    held, handles = [], set()
    for i in range(30):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        handles.add(s.value)
        n_ticks = 1 if i % 2 == 0 else 3
        if i % 2 == 0:  # the tick waits behind 3 ms of someone else's work on its stream
            H.aid_check(H.testaid().cvgs_debug_occupy(8, 64, 0, 3000.0, s))
        for t in range(n_ticks):
            chains, outs, meta = [], [], []
            for k in range(2):
                crops = H.random_crops(3 + (i + t + k) % 4, fw, fh, seed=14000 + 31 * i + 7 * t + k, wmax=200, hmax=200)
                ops, out, _ = _chain(torch, device, fts[k], crops, (64, 128), 3)
                chains.append(ops)
                outs.append(out)
                meta.append((k, crops))
            lowered = [cvgs.lower(ops) for ops in chains]
            arr = cvgs.pack_chains(lowered)
            torch.cuda.current_stream().synchronize()  # (the tensors' fills, on torch's stream, before the tick on the raw stream)
            capi.check(lib.cvgs_execute_many(arr, len(lowered), s))
            held.append((i, outs, meta, lowered, arr))
        assert hip.hipStreamDestroy(s) == 0  # pending work completes; the handle may come back with the next create
    torch.cuda.synchronize()
    for i, outs, meta, _, _ in held:
        for out, (k, crops) in zip(outs, meta):
            H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frames[k], crops, (64, 128), 3), "stream %d" % i)
    assert len(handles) >= 1


def test_stream_release_retires_the_table_ring(oracle, device, lib):
    """cvgs_stream_release (round 6): the call a host makes before hipStreamDestroy.  Ticks with 16-bit sources take the stream's
    pinned table ring (their descriptors do not travel in the kernel arguments); release waits for them, frees the ring and forgets the handle --
    a stream created afterwards (the runtime likes to hand the same handle out again) ticks from a clean state.  A stream the library holds
    nothing for, and a second release, are no-ops."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    fh, fw = 270, 480
    rng = np.random.default_rng(77)
    frames = [rng.integers(0, 65536, size=(fh, fw, 3), dtype=np.uint16) for _ in range(2)]
    fts = [torch.from_numpy(f.view(np.int16)).to(device) for f in frames]  # (torch has no uint16 arithmetic; the bytes are what matters)
    torch.cuda.synchronize()
    handles = set()
    for i in range(24):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        handles.add(s.value)
        chains, outs, meta = [], [], []
        for k in range(2):
            crops = H.random_crops(4 + (i + k) % 3, fw, fh, seed=13000 + 17 * i + k, wmax=200, hmax=200)
            out = torch.full((len(crops), 3 * 64 * 128), -777.0, dtype=torch.float32, device=device)
            g_src = cvgs.GpuMat(fh, fw, cvgs.CV_16UC3, fts[k].data_ptr(), fw * 6, owner=fts[k])
            ops = H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128), 3, src_depth=cvgs.CV_16U)
            chains.append(ops)
            outs.append(out)
            meta.append((k, crops))
        lowered = [cvgs.lower(ops) for ops in chains]
        arr = cvgs.pack_chains(lowered)
        torch.cuda.current_stream().synchronize()
        for _ in range(3):
            capi.check(lib.cvgs_execute_many(arr, len(lowered), s))
        capi.check(lib.cvgs_stream_release(s))  # waits for the three ticks, retires the ring
        for out, (k, crops) in zip(outs, meta):
            ref = np.full((len(crops), 3 * 64 * 128), -777.0, np.float32)
            oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frames[k], cvgs.CV_16UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), (64, 128), 3,
                                                 src_depth=cvgs.CV_16U)))
            H.assert_bit_exact(out.cpu().numpy(), ref, "16-bit tick on stream %d" % i)
        capi.check(lib.cvgs_stream_release(s))  # nothing left: a no-op
        assert hip.hipStreamDestroy(s) == 0
    assert len(handles) < 24, "the runtime never reused a handle: the scenario this test is about did not occur"
    capi.check(lib.cvgs_stream_release(torch.cuda.current_stream().cuda_stream))  # a stream the library holds nothing for
