"""Regression tests for the round-1 review findings, on the GPU through the C-ABI.

* NV12-resize and warp pushes into a CircularTensor (default and mirrored ring) must reach BOTH write targets: the
  tensor is compared with the oracle over more than BATCH updates (the K4 fast path used to drop the ring slot).
* a CircularTensor update whose frame size differs from the tensor's plane size is refused (it used to write out of bounds).
* resize -> fk::Cast (CAST_TRUNC) -> convertTo chains: the fast kernel must not fold a cast it cannot see through.
* handles created on the device stay usable when the caller's current device is left alone (single-GPU box: device 0)."""
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


@pytest.mark.parametrize("mirrored", [False, True])
@pytest.mark.parametrize("order", [cvgs.NewestFirst, cvgs.OldestFirst])
def test_nv12_resize_push_reaches_ring_and_tensor(oracle, order, mirrored):
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 64, 36, 4
    sw, sh = 256, 144
    ct = cvgs.CircularTensor(cvgs.CV_8UC1, cvgs.CV_32FC1, 3, B, order, cvgs.Standard, W, H_, mirrored=mirrored)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, B, order, cvgs.Standard)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    pw = [cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3])]
    for i in range(2 * B + 2):
        surf = H.random_u8((sh + sh // 2, sw, 1), seed=4000 + i)
        st = torch.from_numpy(surf).to(dev)
        g = cvgs.GpuMat(sh, sw, cvgs.CV_8UC1, st.data_ptr(), sw, owner=st)
        rd = cvgs.read_nv12(g, (W, H_), capi.YUV_FULL, capi.BT709, alpha=False)
        lowered = ct.update(s, rd, *pw, ct.write_split(f))
        if i == 0:
            buf = C.create_string_buffer(128)  # the push must really take the K4 fast path (that is what regressed)
            probe = cvgs.lower([rd, *pw, cvgs.split_tensor(f, 16, W, H_, 1)])
            capi.check(ct.lib.cvgs_kernel_name(C.byref(probe.desc), buf, 128))
            assert buf.value.decode().startswith("k4_nv12_resize"), buf.value
        h = cvgs.GpuMat(sh, sw, cvgs.CV_8UC1, surf.ctypes.data, sw, owner=surf)
        oc.update(cvgs.lower([cvgs.read_nv12(h, (W, H_), capi.YUV_FULL, capi.BT709, alpha=False), *pw,
                              cvgs.WriteIOp(capi.WRITE_TENSOR_SPLIT, f, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), "NV12 push %d (mirrored=%s)" % (i, mirrored))
        del lowered
    ct.release()


@pytest.mark.parametrize("mirrored", [False, True])
@pytest.mark.parametrize("mode", ["full_resolution", "letterbox"])
def test_nv12_pointwise_and_letterbox_pushes_reach_ring_and_tensor(oracle, mode, mirrored):
    """The round-2 additions write both targets too: a decoder surface pushed at full resolution (k_pointwise4's 4:2:0 read mode)
    and pushed through an aspect-ratio-preserving resize (K4's windowed instantiation), default and mirrored ring, > BATCH updates."""
    import torch
    dev = torch.device("cuda:0")
    B = 3
    sw, sh = 136, 72
    W, H_ = (sw, sh) if mode == "full_resolution" else (64, 64)
    ct = cvgs.CircularTensor(cvgs.CV_8UC1, cvgs.CV_32FC1, 3, B, cvgs.NewestFirst, cvgs.Standard, W, H_, mirrored=mirrored)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, B, cvgs.NewestFirst, cvgs.Standard)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    pw = [cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]

    def read(g):
        rd = cvgs.read_nv12(g, None if mode == "full_resolution" else (W, H_), capi.YUV_LIMITED, capi.BT709, alpha=False)
        if mode == "letterbox":
            rd.ar = cvgs.PRESERVE_AR
            rd.background = cvgs._scalar([114.0, 114.0, 114.0])
        return rd

    for i in range(2 * B + 1):
        surf = H.random_u8((sh + sh // 2, sw, 1), seed=4500 + i)
        st = torch.from_numpy(surf).to(dev)
        g = cvgs.GpuMat(sh, sw, cvgs.CV_8UC1, st.data_ptr(), sw, owner=st)
        rd = read(g)
        ct.update(s, rd, *pw, ct.write_split(f))
        if i == 0:
            buf = C.create_string_buffer(128)
            probe = cvgs.lower([rd, *pw, cvgs.split_tensor(f, 16, W, H_, 1)])
            capi.check(ct.lib.cvgs_kernel_name(C.byref(probe.desc), buf, 128))
            assert buf.value.decode() == ("pointwise4_nv12" if mode == "full_resolution" else "k4_nv12_resize_mul_sub_div"), buf.value
        h = cvgs.GpuMat(sh, sw, cvgs.CV_8UC1, surf.ctypes.data, sw, owner=surf)
        oc.update(cvgs.lower([read(h), *pw, cvgs.WriteIOp(capi.WRITE_TENSOR_SPLIT, f, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), "%s push %d (mirrored=%s)" % (mode, i, mirrored))
    ct.release()


@pytest.mark.parametrize("mirrored", [False, True])
def test_warp_push_reaches_ring_and_tensor(oracle, mirrored):
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 48, 40, 3
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, cvgs.OldestFirst, cvgs.Standard, W, H_, mirrored=mirrored)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, B, cvgs.OldestFirst, cvgs.Standard)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    m = [[0.9, 0.1, 3.0], [-0.05, 1.1, 2.0]]
    pw = [cvgs.multiply(f, [0.5] * 3)]
    for i in range(2 * B + 1):
        frame = H.random_u8((90, 120, 3), seed=5000 + i)
        ft = torch.from_numpy(frame).to(dev)
        ct.update(s, cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), m, (W, H_)), *pw, ct.write_split(f))
        oc.update(cvgs.lower([cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), m, (W, H_)), *pw,
                              cvgs.WriteIOp(capi.WRITE_TENSOR_SPLIT, f, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), "warp push %d (mirrored=%s)" % (i, mirrored))
    ct.release()


@pytest.mark.parametrize("mirrored", [False, True])
def test_circular_update_refuses_a_frame_of_another_size(mirrored):
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 32, 24, 3
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, cvgs.NewestFirst, cvgs.Standard, W, H_, mirrored=mirrored)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    for (fw, fh) in ((W + 8, H_), (W, H_ + 2), (W // 2, H_ // 2)):
        frame = torch.zeros((fh, fw, 3), dtype=torch.uint8, device=dev)
        with pytest.raises(capi.CvgsError) as e:  # per-pixel read: the frame IS the plane
            ct.update(s, cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3), cvgs.convertTo(cvgs.CV_8UC3, f), ct.write_split(f))
        assert e.value.code == capi.ERR_INVALID
        big = torch.zeros((100, 100, 3), dtype=torch.uint8, device=dev)
        with pytest.raises(capi.CvgsError) as e:  # resize to a size that is not the tensor's
            ct.update(s, cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(big, cvgs.CV_8UC3), (fw, fh)),
                      ct.write_split(f))
        assert e.value.code == capi.ERR_INVALID
    assert ct.updates() == 0
    torch.cuda.synchronize()
    assert not _read_device(ct.data(), ct.nbytes()).any()
    ct.release()


@pytest.mark.parametrize("final", [cvgs.CV_8U, cvgs.CV_16F])
def test_resize_trunc_cast_then_convert_matches_generic_and_oracle(oracle, final):
    """resize -> fk::Cast<float3,int3> -> convertTo<int3, uchar3 | half3>: after the CAST_TRUNC the value is an integer
    (raw bits), so K1 must not fold the trailing cast into its store."""
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((300, 400, 3), seed=77)
    ft = torch.from_numpy(frame).to(dev)
    dst = (80, 60)
    f, i32, o = cvgs.CV_32FC3, cvgs.CV_32SC3, cvgs.make_type(final, 3)
    np_dt = np.uint8 if final == cvgs.CV_8U else np.float16
    t_dt = torch.uint8 if final == cvgs.CV_8U else torch.float16

    def chain(src, out):
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, src, dst), cvgs.cast(f, i32), cvgs.convertTo(i32, o), cvgs.write(o, out)]

    outs = {}
    for flags in (0, capi.CHAIN_FORCE_GENERIC):
        ot = torch.zeros((dst[1], dst[0], 3), dtype=t_dt, device=dev)
        cvgs.executeOperations(torch.cuda.current_stream(), *chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(ot, o)), flags=flags)
        torch.cuda.synchronize()
        outs[flags] = ot.cpu().numpy()
    ref = np.zeros((dst[1], dst[0], 3), np_dt)
    oracle.execute(cvgs.lower(chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), cvgs.GpuMat.from_array(ref, o))))
    assert ref.any()
    H.assert_bit_exact(outs[capi.CHAIN_FORCE_GENERIC], ref, "generic vs oracle")
    H.assert_bit_exact(outs[0], ref, "dispatcher's kernel vs oracle")


def test_large_batches_recycle_the_descriptor_scratch(oracle, device):
    """> 64 planes: descriptors go through the pooled {pinned, device} scratch; many back-to-back launches on two streams
    with DIFFERENT crop lists must each see their own table (a slot may only be recycled after its kernel)."""
    import torch
    frame = H.random_u8((720, 1280, 3), seed=31)
    ft = torch.from_numpy(frame).to(device)
    g_src = cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    n = 70
    jobs = []
    for k in range(12):
        crops = H.random_crops(n, 1280, 720, seed=900 + k)
        out = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=device)
        jobs.append((crops, out))
    torch.cuda.synchronize()
    keep = []
    for k, (crops, out) in enumerate(jobs):
        st = streams[k % 2]
        with torch.cuda.stream(st):
            keep.append(cvgs.executeOperations(st, *H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1))))
    torch.cuda.synchronize()
    for k, (crops, out) in enumerate(jobs):
        ref = np.zeros((n, 3 * 64 * 128), np.float32)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
        H.assert_bit_exact(out.cpu().numpy(), ref, "job %d" % k)


# ---- round-2 advice: the zero-copy descriptor scratch (tables read in place from pinned NON-COHERENT host memory) --------------
def test_recycled_zero_copy_descriptor_slots_never_serve_a_stale_table(oracle):
    """beyond 320 planes the crop table is read by the kernel straight from a pooled pinned host buffer whose
    slots the host rewrites and recycles; that is only correct if a launch never sees the previous table of a recycled slot
    (every dispatch starts with a system-scope acquire).  160 back-to-back launches on ONE stream, no synchronisation in between,
    a DIFFERENT 330-crop list each, the same few pool slots recycled over and over: every plane of every launch against the
    oracle.  (CVGS_SCRATCH_ZEROCOPY=0 selects the staged copy instead.)"""
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((540, 960, 3), seed=77)
    frame_t = torch.from_numpy(frame).to(dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3)
    h_src = cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)
    n, dst, launches = 330, (16, 8), 160
    plane = 3 * dst[0] * dst[1]
    outs = [torch.zeros((n, plane), dtype=torch.float32, device=dev) for _ in range(launches)]
    lists = [H.random_crops(n, 960, 540, seed=5000 + i, wmin=8, wmax=200, hmin=8, hmax=200) for i in range(launches)]
    s = torch.cuda.current_stream()
    torch.cuda.synchronize()
    names = set()
    for i in range(launches):
        ops = H.k1_chain(g_src, lists[i], cvgs.GpuMat.from_tensor(outs[i], cvgs.CV_32FC1), dst, 3)
        names.add(cvgs.kernel_name(*ops))
        cvgs.executeOperations(s, *ops)
    torch.cuda.synchronize()
    assert names == {"k1_u8c3_swap_mul_sub_div"}, names
    for i in range(launches):
        ref = np.zeros((n, plane), np.float32)
        oracle.execute(cvgs.lower(H.k1_chain(h_src, lists[i], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, 3)))
        H.assert_bit_exact(outs[i].cpu().numpy(), ref, "launch %d through a recycled descriptor slot" % i)


def test_descriptor_slots_behind_a_blocked_stream_are_not_recycled(oracle):
    """Round 4: the scratch pool records ONE event per four launches of a hot stream (a hipEventRecord behind every launch kept the next
    kernel of the stream ~4 us behind), so the last launches of a stream may be covered by no event yet.  Such slots are reclaimed only
    when their stream has nothing left to do (hipStreamQuery, 20 ms after the commit).  Stream S: a 300 ms blocker, then THREE 330-crop
    launches (uncovered); stream T meanwhile: 60 launches with other tables, which need slots for 100+ ms.  S's kernels run last and must
    still read THEIR tables."""
    import time
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((540, 960, 3), seed=78)
    frame_t = torch.from_numpy(frame).to(dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3)
    h_src = cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)
    n, dst = 330, (16, 8)
    plane = 3 * dst[0] * dst[1]
    lib = capi.load_library()
    S, T = torch.cuda.Stream(), torch.cuda.Stream()
    s_lists = [H.random_crops(n, 960, 540, seed=7000 + i, wmin=8, wmax=200, hmin=8, hmax=200) for i in range(3)]
    t_lists = [H.random_crops(n, 960, 540, seed=7100 + i, wmin=8, wmax=200, hmin=8, hmax=200) for i in range(60)]
    s_outs = [torch.zeros((n, plane), dtype=torch.float32, device=dev) for _ in s_lists]
    t_outs = [torch.zeros((n, plane), dtype=torch.float32, device=dev) for _ in t_lists]
    torch.cuda.synchronize()
    # the pool grows to a dozen slots first (12 launches queued behind a 20 ms blocker): adding a slot later would allocate pinned memory,
    # which waits for the device -- i.e. for S's blocker
    H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 20000.0, T.cuda_stream))
    for i in range(12):
        cvgs.executeOperations(T, *H.k1_chain(g_src, t_lists[i], cvgs.GpuMat.from_tensor(t_outs[i], cvgs.CV_32FC1), dst, 3))
    torch.cuda.synchronize()
    for o in t_outs:
        o.zero_()
    torch.cuda.synchronize()
    # (lowered up front: building a 330-crop chain in Python takes milliseconds)
    s_low = [cvgs.lower(H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), dst, 3)) for crops, out in zip(s_lists, s_outs)]
    t_low = [cvgs.lower(H.k1_chain(g_src, crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), dst, 3)) for crops, out in zip(t_lists, t_outs)]
    H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 300000.0, S.cuda_stream))
    for lc in s_low:
        capi.check(lib.cvgs_execute(C.byref(lc.desc), S.cuda_stream))
    # T must not share S's hardware queue (the runtime spreads streams over a few): take the first stream a tiny kernel gets through on
    for cand in [T] + [torch.cuda.Stream() for _ in range(6)]:
        p0 = time.perf_counter()
        H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, cand.cuda_stream))
        cand.synchronize()
        if time.perf_counter() - p0 < 0.02:
            T = cand
            break
    else:
        pytest.skip("every stream of this process queues behind the blocked one")
    t0 = time.perf_counter()
    for i, lc in enumerate(t_low):
        capi.check(lib.cvgs_execute(C.byref(lc.desc), T.cuda_stream))
        if i % 10 == 9:
            T.synchronize()
            time.sleep(0.012)  # (past the 20 ms age at which a quiet stream's slots are looked at -- S is not quiet, it is blocked)
    # (T's launches are normally issued within ~100 ms, while S is still blocked; on a box too slow for that the test still checks every
    #  table, only not the blocked-stream scenario -- no timing assertion in a correctness test)
    torch.cuda.synchronize()
    for lists, outs, name in ((s_lists, s_outs, "blocked stream"), (t_lists, t_outs, "busy stream")):
        for i, (crops, out) in enumerate(zip(lists, outs)):
            ref = np.zeros((n, plane), np.float32)
            oracle.execute(cvgs.lower(H.k1_chain(h_src, crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, 3)))
            H.assert_bit_exact(out.cpu().numpy(), ref, "%s, launch %d" % (name, i))
