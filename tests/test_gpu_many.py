"""cvgs_execute_many and cvgs_write_desc.mirrors on the GPU, bit-exact against the oracle AND against one
cvgs_execute per chain.

execute_many: M independent 50-crop chains (distinct frames, crop lists, output tensors) fused into ONE K1 launch
(grid z = chain) -- the launch-batching answer to the 50-crop latency floor; the reference's closest spelling is the
batch sweep of tests/batchresize/test_batchresize_x_split3D.cu:384-392.
mirrors: the exchange step of the sharded batched-crop path (SURVEY.md 8e option 2): the kernel stores its rows into
its own tensor and into every "peer" tensor; here the peers are further tensors on the same device."""
import ctypes as C
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _make(dev, n_chains, crops_per, frame_hw=(1080, 1920), table=False, seed=100, ragged=False, half=False, **kw):
    import torch
    fh, fw = frame_hw
    chains, outs, refs_in, keep = [], [], [], []
    for m in range(n_chains):
        n = crops_per if not ragged else max(1, crops_per - 7 * m)
        frame = H.random_u8((fh, fw, 3), seed=seed + m)
        crops = H.random_crops(n, fw, fh, seed=seed + 50 + m)
        ft = torch.from_numpy(frame).to(dev)
        ot = torch.full((n, 3 * 64 * 128), -777.0, dtype=torch.float16 if half else torch.float32, device=dev)
        g_src = cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3)
        g_out = cvgs.GpuMat.from_tensor(ot, cvgs.CV_16FC1 if half else cvgs.CV_32FC1)
        ops = H.k1_chain(g_src, crops, g_out, half=half, **kw)
        if table:
            tab = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(dev)
            keep.append(tab)
            ops = H.k1_chain(g_src, crops, g_out, table=tab.data_ptr(), half=half, **kw)
        chains.append(ops)
        outs.append(ot)
        refs_in.append((frame, crops))
        keep.append(ft)
    return chains, outs, refs_in, keep


def _oracle(oracle, frame, crops, half=False, **kw):
    ref = np.full((len(crops), 3 * 64 * 128), -777.0, dtype=np.float16 if half else np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops,
                                         cvgs.GpuMat.from_array(ref, cvgs.CV_16FC1 if half else cvgs.CV_32FC1), half=half, **kw)))
    return ref


@pytest.mark.parametrize("n_chains,crops_per,table,ragged", [(2, 50, False, False), (4, 50, False, True), (16, 50, False, False),
                                                              (3, 100, False, False), (4, 30, True, True), (64, 6, False, False)])
def test_execute_many_matches_separate_and_oracle(oracle, device, n_chains, crops_per, table, ragged):
    import torch
    chains, outs, refs_in, keep = _make(device, n_chains, crops_per, table=table, ragged=ragged)
    lowered, arr = cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    many = [o.cpu().numpy() for o in outs]
    for o in outs:
        o.fill_(-777.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    for m, (frame, crops) in enumerate(refs_in):
        H.assert_bit_exact(many[m], outs[m].cpu().numpy(), "execute_many vs separate launches, chain %d" % m)
        H.assert_bit_exact(many[m], _oracle(oracle, frame, crops), "execute_many vs oracle, chain %d" % m)


def test_execute_many_fp16_and_aspect_ratio(oracle, device):
    import torch
    for kw in ({"half": True}, {"ar": cvgs.PRESERVE_AR, "background": [128.0, 128.0, 128.0, 0.0]}, {"used": 3}):
        chains, outs, refs_in, keep = _make(device, 5, 9, seed=300, **kw)
        cvgs.executeMany(torch.cuda.current_stream(), chains)
        torch.cuda.synchronize()
        for m, (frame, crops) in enumerate(refs_in):
            H.assert_bit_exact(outs[m].cpu().numpy(), _oracle(oracle, frame, crops, **kw), "execute_many %r chain %d" % (kw, m))


def test_execute_many_is_one_launch_and_capturable_with_tables(device, lib):
    """With device plane tables the fused launch allocates and copies nothing: it can be captured into a HIP graph."""
    import torch
    chains, outs, refs_in, keep = _make(device, 8, 50, table=True, seed=500)
    lowered = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(lowered)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()  # frames and tensors were written on torch's stream; `side` does not wait for it by itself
    with torch.cuda.stream(side):
        capi.check(lib.cvgs_execute_many(arr, len(lowered), side.cuda_stream))
        torch.cuda.synchronize()
        want = [o.cpu().numpy() for o in outs]
        for o in outs:
            o.fill_(0.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            capi.check(lib.cvgs_execute_many(arr, len(lowered), torch.cuda.current_stream().cuda_stream))
        g.replay()
        torch.cuda.synchronize()
    for m in range(len(outs)):
        H.assert_bit_exact(outs[m].cpu().numpy(), want[m], "captured execute_many, chain %d" % m)
    # host descriptors: up to 1024 planes in all travel in the kernel arguments of the fused launch (round 5) -- capturable as it is
    chains2, outs2, _, keep2 = _make(device, 2, 330, seed=600)
    low2 = [cvgs.lower(ops) for ops in chains2]
    arr2 = cvgs.pack_chains(low2)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        capi.check(lib.cvgs_execute_many(arr2, 2, side.cuda_stream))
        torch.cuda.synchronize()
        want2 = [o.cpu().numpy() for o in outs2]
        for o in outs2:
            o.fill_(0.0)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            capi.check(lib.cvgs_execute_many(arr2, 2, torch.cuda.current_stream().cuda_stream))
        g2.replay()
        torch.cuda.synchronize()
    for m in range(2):
        H.assert_bit_exact(outs2[m].cpu().numpy(), want2[m], "captured host-described execute_many, chain %d" % m)
    # beyond that (4 x 330 planes) the table would be staged, and a chain of 330 planes does not fit the kernel arguments on its own either:
    # refused under capture, loudly (chains of <= 320 planes ARE capturable one by one:
    # test_execute_many_with_host_descriptors_is_capturable_chain_by_chain)
    chains3, outs3, _, keep3 = _make(device, 4, 330, seed=610)
    low3 = [cvgs.lower(ops) for ops in chains3]
    arr3 = cvgs.pack_chains(low3)
    with torch.cuda.stream(side):
        g3 = torch.cuda.CUDAGraph()
        rc = 0
        with torch.cuda.graph(g3):
            rc = lib.cvgs_execute_many(arr3, 4, torch.cuda.current_stream().cuda_stream)
        assert rc == capi.ERR_UNSUPPORTED, lib.cvgs_last_error()


def test_execute_many_mixed_shapes_falls_back_to_one_by_one(oracle, device):
    """Chains that do not share a K1 shape (different target sizes / programs) are executed one by one, same results."""
    import torch
    frame = H.random_u8((480, 640, 3), seed=9)
    ft = torch.from_numpy(frame).to(device)
    crops = H.random_crops(6, 640, 480, seed=3, wmax=200, hmax=300)
    o1 = torch.zeros((6, 3 * 64 * 128), dtype=torch.float32, device=device)
    o2 = torch.zeros((6, 3 * 32 * 32), dtype=torch.float32, device=device)
    g_src = cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3)
    c1 = H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(o1, cvgs.CV_32FC1))
    c2 = H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(o2, cvgs.CV_32FC1), dst=(32, 32), swap=False)
    cvgs.executeMany(torch.cuda.current_stream(), [c1, c2])
    torch.cuda.synchronize()
    H.assert_bit_exact(o1.cpu().numpy(), _oracle(oracle, frame, crops), "mixed many, chain 0")
    ref2 = np.zeros((6, 3 * 32 * 32), np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops,
                                         cvgs.GpuMat.from_array(ref2, cvgs.CV_32FC1), dst=(32, 32), swap=False)))
    H.assert_bit_exact(o2.cpu().numpy(), ref2, "mixed many, chain 1")


def test_execute_many_rejects_bad_input(lib, device):
    assert lib.cvgs_execute_many(None, 1, None) == capi.ERR_INVALID
    chains, outs, _, keep = _make(device, 2, 4, seed=700)
    low = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(low)
    assert lib.cvgs_execute_many(arr, 0, None) == capi.ERR_INVALID
    assert lib.cvgs_execute_many(arr, capi.MAX_CHAINS + 1, None) == capi.ERR_INVALID
    arr[1].read.batch = 0  # a malformed member: nothing may be enqueued, error reported
    assert lib.cvgs_execute_many(arr, 2, None) == capi.ERR_INVALID


# ---- mirrors ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_mirrors,flags,n", [(1, 0, 10), (3, 0, 10), (7, 0, 10), (2, capi.CHAIN_FORCE_GENERIC, 10), (3, 0, 64), (2, 0, 128), (7, 0, 320)])
def test_mirrors_receive_the_same_rows(oracle, device, n_mirrors, flags, n):
    """Rank r's K1 writes rows [r*n, (r+1)*n) of ITS copy of the [G*n, C*H*W] tensor and of every peer's copy: here the
    peers are further tensors on the same device; every copy must hold the oracle's rows, and nothing else is touched."""
    import torch
    world, rank = 3, 1  # n crops per rank: up to 320 travel in K1's kernel arguments (a 512-crop job on 2 GPUs is 256 per rank)
    frame = H.random_u8((720, 1280, 3), seed=21)
    crops = H.random_crops(n, 1280, 720, seed=22)
    ft = torch.from_numpy(frame).to(device)
    tensors = [torch.full((world * n, 3 * 64 * 128), -5.0, dtype=torch.float32, device=device) for _ in range(1 + n_mirrors)]
    mine = tensors[0][rank * n:(rank + 1) * n]
    ops = H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(mine, cvgs.CV_32FC1))
    ops[-1].mirrored_to([t[rank * n:(rank + 1) * n].data_ptr() for t in tensors[1:]])
    name = cvgs.kernel_name(*ops, flags=flags)
    assert name.startswith("generic" if flags else "k1_u8c3"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=flags)
    torch.cuda.synchronize()
    ref = _oracle(oracle, frame, crops)
    for i, t in enumerate(tensors):
        got = t.cpu().numpy()
        H.assert_bit_exact(got[rank * n:(rank + 1) * n], ref, "mirror %d" % i)
        assert (got[:rank * n] == -5.0).all() and (got[(rank + 1) * n:] == -5.0).all(), "mirror %d: rows of other ranks touched" % i


# ---- fused NV12 chains (K4) --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_P010])
@pytest.mark.parametrize("n_cams,crops_per,ragged,dst", [(4, 12, False, (64, 128)), (6, 50, True, (64, 128)), (16, 5, False, (64, 128)),
                                                         (8, 40, False, (64, 128)), (6, 40, False, (100, 126)), (9, 50, True, (64, 128))])
def test_execute_many_nv12_crop_chains(oracle, device, layout, n_cams, crops_per, ragged, dst):
    """The decode-side form of the batched-crop path: every camera hands over an NV12 (or NV21) decoder surface and a crop
    list; cvgs_execute_many turns them into ONE launch of the K4 kernel -- bit-identical to one launch per camera and to
    the oracle.  From 32 Ki wave-rows on (the last three cases) the 8-bit layouts run four rows per wave with 16-byte stores through an
    LDS transpose (round 5); a 100 x 126 target has a ragged second column tile and a last row group of two rows: the row-by-row stores."""
    import torch
    w, h = 640, 360
    dw, dh = dst
    f = cvgs.CV_32FC3
    chains, outs, refs, keep = [], [], [], []
    for cam in range(n_cams):
        n = crops_per if not ragged else max(1, crops_per - 5 * cam)
        p010 = layout == capi.YUV_P010  # 16-bit samples (10-bit codes + garbage low bits)
        s_t = cvgs.CV_16UC1 if p010 else cvgs.CV_8UC1
        surf = (H.random_u16 if p010 else H.random_u8)((h + h // 2, w), seed=800 + cam)
        st = torch.from_numpy(surf.view(np.int16) if p010 else surf).to(device)
        rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in H.random_crops(n, w, h, seed=850 + cam, wmin=4, wmax=300, hmin=4, hmax=300)]
        ot = torch.full((n, 3 * dw * dh), -3.0, dtype=torch.float32, device=device)
        ref = np.full((n, 3 * dw * dh), -3.0, np.float32)

        def chain(wrap_s, wrap_o, out):
            m = wrap_s(surf)
            luma = cvgs.GpuMat(h, w, s_t, m.data, m.step, owner=m.owner)
            return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (dw, dh), capi.YUV_LIMITED, capi.BT709, False, layout=layout),
                    cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]),
                    cvgs.split(f, wrap_o(out), (dw, dh))]

        chains.append(chain(lambda a: cvgs.GpuMat.from_tensor(st, s_t), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), ot))
        oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, s_t), lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
        outs.append(ot)
        refs.append(ref)
        keep.append(st)
    assert cvgs.kernel_name(*chains[0]).startswith("k4_nv12_resize")
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for cam in range(n_cams):
        H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "fused NV12 chains, camera %d" % cam)
        outs[cam].fill_(-3.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    for cam in range(n_cams):
        H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "one launch per camera, camera %d" % cam)


def test_a_host_described_nv12_tick_is_capturable(oracle, device, lib):
    """The decode-side tick (crops of several NV12 surfaces, host descriptors) carries its planes in the kernel arguments since round 5: the
    fused K4 launch is captured as it is and replays (it used to be refused: K4 chains have no device-table form)."""
    import torch
    w, h = 640, 360
    f = cvgs.CV_32FC3
    chains, outs, refs, keep = [], [], [], []
    for cam in range(7):
        surf = H.random_u8((h + h // 2, w), seed=1800 + cam)
        st = torch.from_numpy(surf).to(device)
        rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in H.random_crops(40, w, h, seed=1850 + cam, wmin=4, wmax=300, hmin=4, hmax=300)]
        ot = torch.full((40, 3 * 64 * 128), -3.0, dtype=torch.float32, device=device)
        ref = np.full((40, 3 * 64 * 128), -3.0, np.float32)

        def chain(wrap_s, wrap_o, out):
            m = wrap_s(surf)
            luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
            return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (64, 128), capi.YUV_LIMITED, capi.BT709, False),
                    cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]),
                    cvgs.split(f, wrap_o(out), (64, 128))]

        chains.append(chain(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_8UC1), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), ot))
        oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
        outs.append(ot)
        refs.append(ref)
        keep.append(st)
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
            for cam in range(7):
                H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "captured NV12 tick, replay %d, camera %d" % (rep, cam))


def test_execute_many_nv12_with_a_narrow_crop_falls_back(oracle, device):
    """A 2-pixel-wide crop is not K4's (its chroma window needs 4 bytes): the set is executed one by one, same results."""
    import torch
    w, h = 64, 32
    f = cvgs.CV_32FC3
    surf = H.random_u8((h + h // 2, w), seed=77)
    st = torch.from_numpy(surf).to(device)
    outs, refs, chains = [], [], []
    for rects in ([(0, 0, 16, 8), (2, 2, 2, 4)], [(4, 4, 20, 10), (8, 0, 40, 30)]):
        ot = torch.zeros((2, 3 * 32 * 16), dtype=torch.float32, device=device)
        ref = np.zeros((2, 3 * 32 * 16), np.float32)

        def chain(wrap_s, wrap_o, out):
            m = wrap_s(surf)
            luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
            return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (32, 16), capi.YUV_FULL, capi.BT601, False), cvgs.split(f, wrap_o(out), (32, 16))]

        chains.append(chain(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_8UC1), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), ot))
        oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
        outs.append(ot)
        refs.append(ref)
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for i in range(2):
        H.assert_bit_exact(outs[i].cpu().numpy(), refs[i], "fallback chain %d" % i)


def test_descriptor_scratch_is_thread_safe(oracle, device, lib):
    """Four host threads, each on its own stream with its own frames and crop lists, hammer the two paths that stage
    descriptors in the library's pooled {pinned, device} scratch -- a 100-crop chain (> 64 planes) and a 3-chain
    cvgs_execute_many -- 40 times each, back to back (ctypes releases the GIL).  A slot recycled before its kernel ran, or
    handed to two calls at once, would show up as another thread's table: every output must equal its own oracle result."""
    import threading
    import torch
    jobs, errors = [], []
    for t in range(4):
        big, big_out, big_in, keep1 = _make(device, 1, 100, frame_hw=(720, 1280), seed=3000 + 10 * t)
        many, many_out, many_in, keep2 = _make(device, 3, 20 + t, frame_hw=(720, 1280), seed=3500 + 10 * t)
        low_big = cvgs.lower(big[0])
        low_many = [cvgs.lower(ops) for ops in many]
        arr = cvgs.pack_chains(low_many)
        jobs.append(dict(stream=torch.cuda.Stream(), low_big=low_big, low_many=low_many, arr=arr, big_out=big_out, many_out=many_out,
                         big_in=big_in, many_in=many_in, keep=(keep1, keep2)))
    torch.cuda.synchronize()

    def worker(j):
        s = j["stream"].cuda_stream
        for _ in range(40):
            rc = lib.cvgs_execute(C.byref(j["low_big"].desc), s)
            if rc:
                errors.append(lib.cvgs_last_error())
            rc = lib.cvgs_execute_many(j["arr"], len(j["low_many"]), s)
            if rc:
                errors.append(lib.cvgs_last_error())

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors[:3]
    for t, j in enumerate(jobs):
        frame, crops = j["big_in"][0]
        H.assert_bit_exact(j["big_out"][0].cpu().numpy(), _oracle(oracle, frame, crops), "thread %d, 100-crop chain" % t)
        for m, (frame, crops) in enumerate(j["many_in"]):
            H.assert_bit_exact(j["many_out"][m].cpu().numpy(), _oracle(oracle, frame, crops), "thread %d, fused chain %d" % (t, m))


@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_P010])
@pytest.mark.parametrize("n", [65, 130, 300, 400])
def test_more_than_64_crops_of_a_decoder_surface_stay_on_k4(oracle, device, layout, n):
    """65+ detections of one NV12 / NV21 / P010 surface in one chain: the crop table no longer fits the kernel arguments; K4
    reads it as one segment of its fused-chain form instead of leaving the chain to the interpreted kernel."""
    import torch
    w, h = 1280, 720
    p010 = layout == capi.YUV_P010
    s_t = cvgs.CV_16UC1 if p010 else cvgs.CV_8UC1
    surf = (H.random_u16 if p010 else H.random_u8)((h + h // 2, w), seed=4100 + n)
    st = torch.from_numpy(surf.view(np.int16) if p010 else surf).to(device)
    rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in H.random_crops(n, w, h, seed=4200 + n, wmin=4, wmax=400, hmin=4, hmax=400)]
    f = cvgs.CV_32FC3
    ot = torch.full((n, 3 * 64 * 128), -3.0, dtype=torch.float32, device=device)
    ref = np.full((n, 3 * 64 * 128), -3.0, np.float32)

    def chain(wrap_s, wrap_o, out):
        m = wrap_s(surf)
        luma = cvgs.GpuMat(h, w, s_t, m.data, m.step, owner=m.owner)
        return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (64, 128), capi.YUV_LIMITED, capi.BT709, False, layout=layout),
                cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]),
                cvgs.split(f, wrap_o(out), (64, 128))]

    ops = chain(lambda a: cvgs.GpuMat.from_tensor(st, s_t), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), ot)
    assert cvgs.kernel_name(*ops).startswith("k4_nv12_resize"), cvgs.kernel_name(*ops)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, s_t), lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
    H.assert_bit_exact(ot.cpu().numpy(), ref, "%d crops of a layout-%d surface" % (n, layout))


# ---- round-2 advice: cvgs_execute_many must keep the meaning of n sequential cvgs_execute calls -------------------------------
def test_execute_many_with_overlapping_outputs_keeps_the_sequential_meaning(oracle, device):
    """Two same-shape chains that write the SAME tensor: fused they would race; one by one the second wins."""
    import torch
    frame_a, frame_b = H.random_u8((720, 1280, 3), seed=901), H.random_u8((720, 1280, 3), seed=902)
    crops = H.random_crops(20, 1280, 720, seed=903)
    ta, tb = torch.from_numpy(frame_a).to(device), torch.from_numpy(frame_b).to(device)
    out = torch.zeros((20, 3 * 64 * 128), dtype=torch.float32, device=device)
    g_out = cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)
    chains = [H.k1_chain(cvgs.GpuMat.from_tensor(ta, cvgs.CV_8UC3), crops, g_out), H.k1_chain(cvgs.GpuMat.from_tensor(tb, cvgs.CV_8UC3), crops, g_out)]
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    H.assert_bit_exact(out.cpu().numpy(), _oracle(oracle, frame_b, crops), "the later chain's result stands")


def test_execute_many_with_host_descriptors_is_capturable_chain_by_chain(oracle, device):
    """Host descriptors under stream capture: the fused launch would stage a table (not capturable); the call falls back to one
    launch per chain, whose <= 64 planes travel in kernel arguments."""
    import torch
    chains, outs, refs_in, keep = _make(device, 3, 40)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for o in outs:
        o.fill_(-777.0)
    g.replay()
    torch.cuda.synchronize()
    for o, (frame, crops) in zip(outs, refs_in):
        H.assert_bit_exact(o.cpu().numpy(), _oracle(oracle, frame, crops), "captured execute_many, host descriptors")


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVGS_FUZZ_K4_TICKS_N", "8"))))
def test_random_nv12_ticks(oracle, device, seed):
    """Random decode-side ticks: 2-14 surfaces of a random 4:2:0 layout (NV12 / NV21 / P010), 1-60 crops each, a random target size (ragged column
    tiles, row counts that are not multiples of four): one-row and four-row waves, inline descriptors and the table ring, bit-exact against the
    oracle.  CVGS_FUZZ_K4_TICKS_N=300 for a long hunt."""
    import torch
    rng = np.random.default_rng(77000 + seed)
    layout = [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_P010][int(rng.integers(3))]
    p010 = layout == capi.YUV_P010
    dw, dh = int(rng.choice([16, 48, 64, 100, 128, 200])), int(rng.choice([8, 30, 64, 126, 128, 131]))
    w, h = 640, 360
    f = cvgs.CV_32FC3
    s_t = cvgs.CV_16UC1 if p010 else cvgs.CV_8UC1
    chains, outs, refs, keep = [], [], [], []
    for cam in range(int(rng.integers(2, 15))):
        n = int(rng.integers(1, 61))
        surf = (H.random_u16 if p010 else H.random_u8)((h + h // 2, w), seed=int(rng.integers(1 << 30)))
        st = torch.from_numpy(surf.view(np.int16) if p010 else surf).to(device)
        rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in H.random_crops(n, w, h, seed=int(rng.integers(1 << 30)), wmin=4, wmax=300, hmin=4, hmax=300)]
        ot = torch.full((n, 3 * dw * dh), -3.0, dtype=torch.float32, device=device)
        ref = np.full((n, 3 * dw * dh), -3.0, np.float32)

        def chain(wrap_s, wrap_o, out):
            m = wrap_s(surf)
            luma = cvgs.GpuMat(h, w, s_t, m.data, m.step, owner=m.owner)
            return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], (dw, dh), capi.YUV_LIMITED, capi.BT709, False, layout=layout),
                    cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]),
                    cvgs.split(f, wrap_o(out), (dw, dh))]

        chains.append(chain(lambda a: cvgs.GpuMat.from_tensor(st, s_t), lambda o: cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), ot))
        oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, s_t), lambda o: cvgs.GpuMat.from_array(o, cvgs.CV_32FC1), ref)))
        outs.append(ot)
        refs.append(ref)
        keep.append(st)
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for cam in range(len(outs)):
        H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "random NV12 tick %d (layout %d, %dx%d), camera %d" % (seed, layout, dw, dh, cam))


# ---- round 6 (VERDICT r5 "what's weak" #6): the independence check covers device-table chains -------------------------------------------
def _captured_kernel_nodes(lib, arr, n, device):
    """How many kernel launches does ONE cvgs_execute_many call enqueue?  Captured on a side stream with the HIP runtime's own capture calls
    and counted with hipGraphGetNodes (device tables: every path of the call is capturable)."""
    import torch
    hip = C.CDLL(None)  # libamdhip64 came in with torch; its symbols are global in this process
    for name in ("hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphGetNodes", "hipGraphDestroy"):
        if not hasattr(hip, name):
            import glob
            hip = C.CDLL(sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")))[0])
            break
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    graph = C.c_void_p()
    assert hip.hipStreamBeginCapture(C.c_void_p(side.cuda_stream), 0) == 0  # hipStreamCaptureModeGlobal
    rc = lib.cvgs_execute_many(arr, n, side.cuda_stream)
    assert hip.hipStreamEndCapture(C.c_void_p(side.cuda_stream), C.byref(graph)) == 0
    capi.check(rc)
    count = C.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, C.byref(count)) == 0
    hip.hipGraphDestroy(graph)
    return count.value


def test_device_table_chains_are_checked_for_independence(oracle, device, lib):
    """Two device-table chains of one shape where chain B READS the tensor chain A WRITES (a tensor reinterpreted as a u8 frame): fused they
    would run concurrently and B would read stale bytes.  With its source range stated (read.table_src_lo / _hi, which the Python facade
    fills from cvgs_plane_table_hull) the call runs them one by one, in order; the same without any statement (safe default); a caller that
    VOUCHES for independent chains gets the one fused launch."""
    import torch
    n, fw, fh = 12, 640, 384
    frame = H.random_u8((fh, fw, 3), seed=1201)
    ft = torch.from_numpy(frame).to(device)
    out_a = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=device)  # 12 x 98,304 B = 1,179,648 B
    # chain B's "frame": the first 384 x 640 x 3 bytes of A's output tensor, read as u8 pixels
    alias = out_a.view(torch.uint8).reshape(-1)[: fh * fw * 3].view(fh, fw, 3)
    out_b = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=device)
    crops_a, crops_b = H.random_crops(n, fw, fh, seed=1202, wmax=300, hmax=300), H.random_crops(n, fw, fh, seed=1203, wmax=300, hmax=300)

    def chains(mode):
        res, keep = [], []
        for src_t, crops, out in ((ft, crops_a, out_a), (alias, crops_b, out_b)):
            g_src, g_out = cvgs.GpuMat.from_tensor(src_t, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)
            ops = H.k1_chain(g_src, crops, g_out)
            tab = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(device)
            keep.append(tab)
            ops = H.k1_chain(g_src, crops, g_out, table=tab.data_ptr())
            if mode == "unstated":
                ops[0].mats = None  # no host views at hand: the facade cannot compute a hull
            elif mode == "vouched":
                ops[0].table_vouched = True
            res.append(cvgs.lower(ops))
        return res, keep

    # what n sequential cvgs_execute calls give: B resizes A's OUTPUT BYTES
    ref_a = np.zeros((n, 3 * 64 * 128), np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops_a, cvgs.GpuMat.from_array(ref_a, cvgs.CV_32FC1))))
    frame_b = np.ascontiguousarray(ref_a.view(np.uint8).reshape(-1)[: fh * fw * 3].reshape(fh, fw, 3))
    ref_b = np.zeros((n, 3 * 64 * 128), np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame_b, cvgs.CV_8UC3), crops_b, cvgs.GpuMat.from_array(ref_b, cvgs.CV_32FC1))))
    for mode in ("stated", "unstated"):
        low, keep = chains(mode)
        if mode == "stated":
            assert low[1].desc.read.table_src_lo == alias.data_ptr() + 0 or low[1].desc.read.table_src_lo >= alias.data_ptr()
            assert low[1].desc.read.table_src_hi <= alias.data_ptr() + fh * fw * 3
        else:
            assert not low[1].desc.read.table_src_lo and not (low[1].desc.read.flags & capi.READ_FLAG_TABLE_SOURCES_VOUCHED)
        arr = cvgs.pack_chains(low)
        for _ in range(3):  # a race would not lose every time; the sequential meaning never loses
            out_a.zero_()
            out_b.zero_()
            torch.cuda.synchronize()
            capi.check(lib.cvgs_execute_many(arr, 2, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            H.assert_bit_exact(out_a.cpu().numpy(), ref_a, "chain A (%s)" % mode)
            H.assert_bit_exact(out_b.cpu().numpy(), ref_b, "chain B reads what chain A wrote (%s)" % mode)
        assert _captured_kernel_nodes(lib, arr, 2, device) == 2, "dependent device-table chains must run one by one (%s)" % mode
    # independent chains (distinct frames, distinct tensors): stated hulls -> ONE fused launch; unstated -> one by one; vouched -> ONE launch
    ch2, outs2, refs2, keep2 = _make(device, 3, 20, table=True, seed=1300)
    for mode, want_nodes in (("stated", 1), ("unstated", 3), ("vouched", 1)):
        low = []
        for ops in ch2:
            rd = ops[0]
            saved = rd.mats
            if mode == "unstated":
                rd.mats = None
            rd.table_vouched = mode == "vouched"
            if mode == "vouched":
                rd.table_hull = None
                rd.mats = None
            low.append(cvgs.lower(ops))
            rd.mats = saved
            rd.table_vouched = False
        arr = cvgs.pack_chains(low)
        for o in outs2:
            o.fill_(-777.0)
        torch.cuda.synchronize()
        capi.check(lib.cvgs_execute_many(arr, 3, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        for m, (fr, crops) in enumerate(refs2):
            H.assert_bit_exact(outs2[m].cpu().numpy(), _oracle(oracle, fr, crops), "independent device-table chains (%s), chain %d" % (mode, m))
        assert _captured_kernel_nodes(lib, arr, 3, device) == want_nodes, mode


# ---- round 6: ticks of POINTWISE chains (the reference's batched per-pixel chains, tests/batchread/test_batchread_x_write3D.cu:92-96) ------------------
def _pointwise_tick(device, n_chains, batch, cn, size=(60, 120), kind="write3d", seed=2000, ragged=False, used=None, prog="norm"):
    import torch
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    w, h = size
    chains, outs, meta, keep = [], [], [], []
    for m in range(n_chains):
        b = batch if not ragged else max(1, batch - 3 * m)
        frame = H.random_u8((400, 600, cn), seed=seed + m)
        ft = torch.from_numpy(frame).to(device)
        rects = [((7 * i + 3 * m) % (600 - w), (5 * i + m) % (400 - h), w, h) for i in range(b)]
        out = torch.full((b, w * h * cn), -777.0, dtype=torch.float32, device=device)

        def build(mat, o, rects=rects, b=b):
            ops = [cvgs.ReadIOp(capi.READ_PIXEL, u, [mat.roi(*r) for r in rects], b if used is None else min(used, b))]
            if prog == "norm":
                ops += [cvgs.convertTo(u, f, 0.3), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn])]
            elif prog == "cast":
                ops += [cvgs.convertTo(u, f)]
            else:  # an interpreted arithmetic program
                ops += [cvgs.convertTo(u, f), cvgs.add(f, [1.5, 2.5, 3.5, 4.5][:cn]), cvgs.multiply(f, [0.5] * cn)]
            ops.append(cvgs.write(f, o, (w, h)) if kind == "write3d" else cvgs.split(f, o, (w, h)))
            return ops
        chains.append(build(cvgs.GpuMat.from_tensor(ft, u), cvgs.GpuMat.from_tensor(out, f if kind == "write3d" else cvgs.CV_32FC1)))
        outs.append(out)
        meta.append((frame, build, b))
        keep.append(ft)
    return chains, outs, meta, keep


@pytest.mark.parametrize("cn,kind,prog,ragged,used", [(3, "write3d", "norm", False, None), (4, "write3d", "norm", True, None), (1, "split", "cast", False, None),
                                                        (2, "write3d", "interp", True, None), (3, "split", "norm", False, 5), (4, "split", "interp", False, None)])
def test_pointwise_chains_fuse_into_one_launch(oracle, device, lib, cn, kind, prog, ragged, used):
    """16 batched per-pixel chains of one shape in ONE launch (k_pointwise4_many: the planes of all chains in the kernel arguments, grid z =
    chain x max_batch + plane): bit-identical to one cvgs_execute per chain and to the oracle; ONE captured kernel node."""
    import torch
    chains, outs, meta, keep = _pointwise_tick(device, 16, 12, cn, kind=kind, prog=prog, ragged=ragged, used=used)
    lowered = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(lowered)
    torch.cuda.synchronize()
    capi.check(lib.cvgs_execute_many(arr, len(lowered), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    many = [o.cpu().numpy() for o in outs]
    for o in outs:
        o.fill_(-777.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    f = cvgs.make_type(cvgs.CV_32F, cn)
    for m, (frame, build, b) in enumerate(meta):
        H.assert_bit_exact(many[m], outs[m].cpu().numpy(), "pointwise tick vs separate launches, chain %d" % m)
        ref = np.full((b, 60 * 120 * cn), -777.0, np.float32)
        oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), cvgs.GpuMat.from_array(ref, f if kind == "write3d" else cvgs.CV_32FC1))))
        H.assert_bit_exact(many[m], ref, "pointwise tick vs oracle, chain %d" % m)
    assert _captured_kernel_nodes(lib, arr, len(lowered), device) == 1


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("size", [(4, 4), (8, 3), (12, 5), (16, 7), (32, 9), (60, 17), (64, 6), (62, 9), (68, 5), (100, 3), (124, 2), (128, 7), (132, 5)])
def test_narrow_dense_planes_of_every_width(oracle, device, lib, cn, size):
    """Planes at most 64 / 128 pixels wide put 4 / 2 rows on a wave; with dense rows (the tensor of cvGS::write / fk::TensorWrite) the wave's rows leave
    as ONE span of 16-byte chunks through LDS (k_pointwise_body.hpp, round 6).  Widths around both limits, heights that are not a multiple of the
    rows per wave (the last wave stores directly), a width that is not a multiple of 4: as a tick AND one by one, against the oracle."""
    import torch
    w, h = size
    chains, outs, meta, keep = _pointwise_tick(device, 3, 7, cn, size=size, seed=2400 + w, ragged=True)
    lowered = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(lowered)
    torch.cuda.synchronize()
    capi.check(lib.cvgs_execute_many(arr, len(lowered), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    many = [o.cpu().numpy() for o in outs]
    for o in outs:
        o.fill_(-777.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    f = cvgs.make_type(cvgs.CV_32F, cn)
    for m, (frame, build, b) in enumerate(meta):
        ref = np.full((b, w * h * cn), -777.0, np.float32)
        oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), cvgs.GpuMat.from_array(ref, f))))
        H.assert_bit_exact(outs[m].cpu().numpy(), ref, "narrow plane %dx%d C%d one by one, chain %d" % (w, h, cn, m))
        H.assert_bit_exact(many[m], ref, "narrow plane %dx%d C%d as a tick, chain %d" % (w, h, cn, m))


def test_pointwise_ticks_that_do_not_fuse_keep_their_meaning(oracle, device, lib):
    """Chains of different plane sizes, and chains that alias (B reads what A writes), run one by one -- same results as separate calls."""
    import torch
    a_chains, a_outs, a_meta, keep_a = _pointwise_tick(device, 2, 6, 3, size=(60, 120), seed=2100)
    b_chains, b_outs, b_meta, keep_b = _pointwise_tick(device, 1, 6, 3, size=(40, 80), seed=2200)
    # (the write extents differ, so same_shape already says no; the call must still be right)
    chains = a_chains + b_chains
    outs = a_outs + b_outs
    lowered = [cvgs.lower(ops) for ops in chains]
    arr = cvgs.pack_chains(lowered)
    torch.cuda.synchronize()
    capi.check(lib.cvgs_execute_many(arr, 3, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = [o.cpu().numpy() for o in outs]
    for o in outs:
        o.fill_(-777.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    for m in range(3):
        H.assert_bit_exact(got[m], outs[m].cpu().numpy(), "mixed plane sizes, chain %d" % m)
    assert _captured_kernel_nodes(lib, arr, 3, device) == 3
    # two chains writing the SAME tensor: the later one's result stands
    c2, o2, m2, k2 = _pointwise_tick(device, 2, 6, 3, seed=2300)
    frame1, build1, b1 = m2[1]
    ft1 = k2[1]
    ops1 = build1(cvgs.GpuMat.from_tensor(ft1, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(o2[0], cvgs.CV_32FC3))
    low = [cvgs.lower(c2[0]), cvgs.lower(ops1)]
    arr2 = cvgs.pack_chains(low)
    torch.cuda.synchronize()
    capi.check(lib.cvgs_execute_many(arr2, 2, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ref = np.full((b1, 60 * 120 * 3), -777.0, np.float32)
    oracle.execute(cvgs.lower(build1(cvgs.GpuMat.from_array(frame1, cvgs.CV_8UC3), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC3))))
    H.assert_bit_exact(o2[0].cpu().numpy(), ref, "aliased pointwise chains: the later chain's result stands")
    assert _captured_kernel_nodes(lib, arr2, 2, device) == 2
