"""Stream-ordered submission to the descriptor queue (cvgs_queue_submit_on, ABI 5): the reference's contract for
cvGS::executeOperations(stream, iops...) -- "asynchronous on the given stream", include/cvGPUSpeedup.cuh:464-473 -- on the queue.
A producer kernel on stream S rewrites the frame, the submit follows WITHOUT any host synchronisation, a consumer kernel on S reads the
tensor; every iteration is compared with the CPU oracle bit for bit (on the device, so that no host round trip orders anything).
Also: the deferred-wait form (several batches of one stream in flight), several streams on one queue, the hybrid latency policy, a gate
that stays closed longer than the stall watchdog, and recovery after the watchdog has fired (cvgs_queue_recover)."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu

DST, CN = (64, 128), 3


@pytest.fixture()
def torch_dev():
    import torch
    return torch, torch.device("cuda:0")


def oracle_out(oracle, frame_np, crops, n):
    ref = np.full((n, CN * DST[0] * DST[1]), -777.0, dtype=np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame_np, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), DST, CN)))
    return ref


class Camera:
    """One stream's world: a frame buffer the producer rewrites, a tensor the consumer reads, a pool of source pictures with their
    oracle results, and a device-side mismatch counter (everything allocated up front: an allocation while a server is alive
    synchronises the device)."""

    def __init__(self, torch, dev, oracle, seed, n_crops=12, pool=8, wh=(960, 540)):
        w, h = wh
        self.torch = torch
        self.crops = H.random_crops(n_crops, w, h, wmax=300, hmax=400, seed=seed)
        self.pics = [H.random_u8((h, w, 3), seed=seed * 100 + i) for i in range(pool)]
        self.pool = [torch.from_numpy(p).to(dev) for p in self.pics]
        self.refs = [torch.from_numpy(oracle_out(oracle, p, self.crops, n_crops)).to(dev).view(torch.int32) for p in self.pics]
        self.frame = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        self.out = torch.full((n_crops, CN * DST[0] * DST[1]), -777.0, dtype=torch.float32, device=dev)
        self.bad = torch.zeros((), dtype=torch.int64, device=dev)
        self.tmp = torch.zeros((n_crops, CN * DST[0] * DST[1]), dtype=torch.bool, device=dev)
        self.tmp_sum = torch.zeros((), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream()
        ops = H.k1_chain(cvgs.GpuMat.from_tensor(self.frame, cvgs.CV_8UC3), self.crops, cvgs.GpuMat.from_tensor(self.out, cvgs.CV_32FC1), DST, CN)
        self.lowered = cvgs.lower(ops)

    def produce(self, i):      # a kernel on the stream REWRITES the frame
        self.frame.copy_(self.pool[i % len(self.pool)], non_blocking=True)

    def consume(self, i):      # kernels on the stream READ the tensor: mismatches against the oracle's bits are counted on the device
        torch = self.torch
        torch.ne(self.out.view(torch.int32), self.refs[i % len(self.refs)], out=self.tmp)
        torch.sum(self.tmp, dim=(0, 1), out=self.tmp_sum)
        self.bad.add_(self.tmp_sum)
        self.out.fill_(-1.0)   # and poison it: a batch that was skipped, or read early, cannot pass


def warm(torch, cam):
    with torch.cuda.stream(cam.stream):
        cam.produce(0)
        cam.consume(0)
    cam.stream.synchronize()
    cam.bad.zero_()
    torch.cuda.synchronize()


def test_producer_submit_consumer_on_one_stream_1000_iterations(oracle, torch_dev):
    """VERDICT r3 #2's acceptance test: producer kernel -> executeOperations on the queue -> consumer kernel, all on ONE stream, no
    synchronisation of any kind inside the loop, a different picture in the SAME frame buffer every iteration."""
    torch, dev = torch_dev
    cam = Camera(torch, dev, oracle, seed=5)
    warm(torch, cam)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        with torch.cuda.stream(cam.stream):
            for i in range(1000):
                cam.produce(i)
                t = q.submit_lowered_on(cam.stream, cam.lowered)
                assert t != cvgs.Queue.TICKET_DIRECT
                cam.consume(i)
        cam.stream.synchronize()
        st = q.stats()
        assert st["error"] == 0 and st["submitted"] == 1000, st
        assert int(cam.bad.item()) == 0, "%d mismatching elements over 1000 stream-ordered batches" % int(cam.bad.item())
    finally:
        q.destroy()


def test_four_streams_share_one_queue(oracle, torch_dev):
    """Four cameras, four streams, one queue: each stream's batches are ordered behind its own producer and in front of its own
    consumer; the batches of different streams overlap on the server."""
    torch, dev = torch_dev
    cams = [Camera(torch, dev, oracle, seed=20 + k, n_crops=8 + 3 * k) for k in range(4)]
    for c in cams:
        warm(torch, c)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        for i in range(250):
            for c in cams:
                with torch.cuda.stream(c.stream):
                    c.produce(i)
                    q.submit_lowered_on(c.stream, c.lowered)
                    c.consume(i)
        for c in cams:
            c.stream.synchronize()
        st = q.stats()
        assert st["error"] == 0 and st["submitted"] == 1000, st
        assert [int(c.bad.item()) for c in cams] == [0, 0, 0, 0]
    finally:
        q.destroy()


def test_a_tick_of_six_cameras_behind_one_gate(oracle, torch_dev):
    """cvgs_queue_submit_many_on: the six pictures of one tick are rewritten by producer kernels on ONE stream, their six chains go
    behind ONE gate kernel, six consumers follow on the stream; 150 ticks, no synchronisation inside the loop."""
    torch, dev = torch_dev
    s = torch.cuda.Stream()
    cams = [Camera(torch, dev, oracle, seed=120 + k, n_crops=6 + 5 * k, pool=4) for k in range(6)]
    for c in cams:
        c.stream = s
        warm(torch, c)
    ptrs = cvgs.Queue.chain_pointers([c.lowered for c in cams])
    q = cvgs.Queue(idle_us=5000.0)
    try:
        with torch.cuda.stream(s):
            for i in range(150):
                for c in cams:
                    c.produce(i)
                q.submit_many_on(s, ptrs, len(cams))
                for c in cams:
                    c.consume(i)
        s.synchronize()
        st = q.stats()
        assert st["error"] == 0 and st["submitted"] == 150 * 6, st
        assert sum(int(c.bad.item()) for c in cams) == 0
    finally:
        q.destroy()


def test_deferred_wait_keeps_several_batches_of_one_stream_in_flight(oracle, torch_dev):
    """CVGS_QUEUE_SUBMIT_DEFER_WAIT: the gate still orders every batch behind its producer, but the stream is not held on the batch;
    the consumer is ordered by cvgs_queue_stream_wait.  Eight frame buffers / tensors in rotation, one stream."""
    torch, dev = torch_dev
    cams = [Camera(torch, dev, oracle, seed=40 + k, pool=4) for k in range(8)]
    s = torch.cuda.Stream()
    for c in cams:
        c.stream = s
        warm(torch, c)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        tickets = [None] * 8
        with torch.cuda.stream(s):
            for i in range(400):
                c = cams[i % 8]
                if tickets[i % 8] is not None:        # the slot's previous batch: consumer behind its ticket, then the buffer is free
                    q.stream_wait(tickets[i % 8], s)
                    c.consume(i // 8 - 1)
                c.produce(i // 8)
                tickets[i % 8] = q.submit_lowered_on(s, c.lowered, cvgs.Queue.DEFER_WAIT)
            for k in range(8):
                i = 400 + k
                q.stream_wait(tickets[i % 8], s)
                cams[i % 8].consume(i // 8 - 1)
        s.synchronize()
        assert q.stats()["error"] == 0
        assert sum(int(c.bad.item()) for c in cams) == 0
    finally:
        q.destroy()


def test_hybrid_policy_takes_the_direct_launch_for_a_lone_stream(oracle, torch_dev):
    """A strictly ordered stream alone on its queue has nothing to overlap with: CVGS_QUEUE_SUBMIT_HYBRID launches the chain directly on
    the stream (the ticket says so) -- same bits, half the latency.  With a second stream's batch open the server takes it -- once the
    group reaches the policy's minimum (MIN_GROUP(1) here: every call; the default 8 keeps single calls as launches)."""
    torch, dev = torch_dev
    a, b = Camera(torch, dev, oracle, seed=60), Camera(torch, dev, oracle, seed=61)
    warm(torch, a)
    warm(torch, b)
    q = cvgs.Queue(idle_us=5000.0)
    lib = capi.load_library()
    try:
        with torch.cuda.stream(a.stream):
            for i in range(20):
                a.produce(i)
                assert q.submit_lowered_on(a.stream, a.lowered, cvgs.Queue.HYBRID) == cvgs.Queue.TICKET_DIRECT
                a.consume(i)
        a.stream.synchronize()
        assert int(a.bad.item()) == 0 and q.stats()["submitted"] == 0
        # stream b's batch is held open by a slow producer (20 ms); stream a's submit now has something to overlap with
        with torch.cuda.stream(b.stream):
            H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 20000.0, b.stream.cuda_stream))
            b.produce(3)
            tb = q.submit_lowered_on(b.stream, b.lowered)
            b.consume(3)
        with torch.cuda.stream(a.stream):
            a.produce(5)
            td = q.submit_lowered_on(a.stream, a.lowered, cvgs.Queue.HYBRID)                             # default minimum (8): a launch
            a.consume(5)
            a.produce(6)
            ta = q.submit_lowered_on(a.stream, a.lowered, cvgs.Queue.HYBRID | cvgs.Queue.MIN_GROUP(1))  # every call may go to the server
            a.consume(6)
        assert tb != cvgs.Queue.TICKET_DIRECT and ta != cvgs.Queue.TICKET_DIRECT and td == cvgs.Queue.TICKET_DIRECT
        a.stream.synchronize()
        b.stream.synchronize()
        assert int(a.bad.item()) == 0 and int(b.bad.item()) == 0 and q.stats()["error"] == 0
    finally:
        q.destroy()


def test_a_gate_closed_longer_than_the_stall_limit_is_waiting_not_a_stall(oracle, torch_dev):
    """The producer takes 400 ms (the watchdog's limit is 250 ms): the batch must not complete before the producer has finished, the
    watchdog must not fire, and the result must be computed from the producer's pixels."""
    torch, dev = torch_dev
    cam = Camera(torch, dev, oracle, seed=70)
    warm(torch, cam)
    q = cvgs.Queue(idle_us=5000.0)
    lib = capi.load_library()
    try:
        with torch.cuda.stream(cam.stream):
            H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 400000.0, cam.stream.cuda_stream))
            cam.produce(2)
            q.submit_lowered_on(cam.stream, cam.lowered)
            cam.consume(2)
        time.sleep(0.1)
        st = q.stats()
        assert st["submitted"] == 1 and st["completed"] == 0 and st["error"] == 0, st  # published, its gate still closed
        cam.stream.synchronize()
        st = q.stats()
        assert st["completed"] == 1 and st["error"] == 0, st
        assert int(cam.bad.item()) == 0
    finally:
        q.destroy()


def test_the_watchdog_fires_and_the_queue_recovers(oracle, torch_dev, monkeypatch):
    """A foreign kernel keeps half of the CUs' LDS for 300 ms, so a third of a three-per-CU server's workgroups cannot become resident; with a 50 ms
    stall limit the watchdog reports the batch (CVGS_ERR_HIP on the wait -- not a hang).  cvgs_queue_recover then resets the queue:
    the lost batch is reported through its ticket, later submits are served by a fresh server, bit-exact."""
    torch, dev = torch_dev
    monkeypatch.setenv("CVGS_QUEUE_STALL_MS", "50")
    frame = H.random_u8((540, 960, 3), seed=81)
    crops = H.random_crops(40, 960, 540, wmax=300, hmax=400, seed=82)
    ref = oracle_out(oracle, frame, crops, 40)
    frame_t = torch.from_numpy(frame).to(dev)
    out_t = torch.full((40, CN * DST[0] * DST[1]), -777.0, dtype=torch.float32, device=dev)
    lowered = cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), DST, CN))
    hog = torch.cuda.Stream()
    lib = capi.load_library()
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    q = cvgs.Queue(flags=(3 * cus - 1) << 16)                   # three workgroups per CU: more than the half-occupied chip can hold (4 per free CU)
    try:
        q.wait(q.submit_lowered(lowered))                      # a healthy round first
        H.assert_bit_exact(out_t.cpu().numpy(), ref, "before the stall")
        torch.cuda.synchronize()                                # the server has retired
        H.aid_check(H.testaid().cvgs_debug_occupy(cus // 2, 64, 150 * 1024, 300000.0, hog.cuda_stream))
        time.sleep(0.01)
        t = q.submit_lowered(lowered)
        with pytest.raises(capi.CvgsError):
            q.wait(t, timeout_s=5.0)
        assert q.stats()["error"] == 1
        with pytest.raises(capi.CvgsError):                    # dead until recovered
            q.submit_lowered(lowered)
        lost = q.recover()                                      # (waits for the failed server: its late workgroups run when the hog ends)
        assert q.stats()["error"] == 0
        if lost:                                                # a lost batch stays reported through its ticket
            with pytest.raises(capi.CvgsError):
                q.wait(t, timeout_s=1.0)
        else:                                                   # the late workgroups finished it after the watchdog had fired: complete, not lost
            q.wait(t, timeout_s=1.0)
            H.assert_bit_exact(out_t.cpu().numpy(), ref, "finished late")
        hog.synchronize()
        for _ in range(3):
            out_t.fill_(-5.0)
            torch.cuda.current_stream().synchronize()
            q.wait(q.submit_lowered(lowered))
            H.assert_bit_exact(out_t.cpu().numpy(), ref, "after recovery")
        assert q.recover() == 0                                 # a no-op on a healthy queue
    finally:
        q.destroy()


def test_fewer_than_sixteen_workers_are_clamped(oracle, torch_dev):
    """a task of residue class T % 16 is only drawn by workers of that class; G = 1 (4 workers) would leave 12 classes
    unserved.  The create call clamps G to 4 workgroups; a 12-crop batch (>= 5 tasks) completes."""
    torch, dev = torch_dev
    frame = H.random_u8((540, 960, 3), seed=91)
    crops = H.random_crops(12, 960, 540, wmax=300, hmax=400, seed=92)
    frame_t = torch.from_numpy(frame).to(dev)
    out_t = torch.full((12, CN * DST[0] * DST[1]), -777.0, dtype=torch.float32, device=dev)
    lowered = cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1), DST, CN))
    torch.cuda.synchronize()
    q = cvgs.Queue(flags=1 << 16)  # bits 16..27: worker workgroups = 1
    try:
        assert q.stats()["workgroups"] == 4
        q.wait(q.submit_lowered(lowered), timeout_s=5.0)
        H.assert_bit_exact(out_t.cpu().numpy(), oracle_out(oracle, frame, crops, 12), "G clamped to 4")
    finally:
        q.destroy()


def test_submit_on_refuses_a_capturing_stream_and_hybrid_captures_the_launch(oracle, torch_dev):
    torch, dev = torch_dev
    cam = Camera(torch, dev, oracle, seed=95)
    warm(torch, cam)
    q = cvgs.Queue()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = torch.cuda.current_stream()
            with pytest.raises(capi.CvgsError):
                q.submit_lowered_on(s, cam.lowered)
            assert q.submit_lowered_on(s, cam.lowered, cvgs.Queue.HYBRID) == cvgs.Queue.TICKET_DIRECT
        cam.frame.copy_(cam.pool[1])
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert bool(torch.equal(cam.out.view(torch.int32), cam.refs[1]))
    finally:
        q.destroy()


def test_attach_queue_keeps_the_call_shape(oracle, torch_dev):
    """cvgs.attachQueue(stream, queue): the reference's call -- executeOperations(stream, iops...) -- unchanged, two attached streams."""
    torch, dev = torch_dev
    cams = [Camera(torch, dev, oracle, seed=130 + k) for k in range(2)]
    for c in cams:
        warm(torch, c)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        for c in cams:
            cvgs.attachQueue(c.stream, q, minGroup=1)  # (the default policy would launch these single calls: same bits, no server)
        for i in range(200):
            for c in cams:
                with torch.cuda.stream(c.stream):
                    c.produce(i)
                    ops = H.k1_chain(cvgs.GpuMat.from_tensor(c.frame, cvgs.CV_8UC3), c.crops, cvgs.GpuMat.from_tensor(c.out, cvgs.CV_32FC1), DST, CN)
                    cvgs.executeOperations(c.stream, *ops)
                    c.consume(i)
        for c in cams:
            c.stream.synchronize()
            cvgs.detachQueue(c.stream)
        assert q.stats()["error"] == 0
        assert [int(c.bad.item()) for c in cams] == [0, 0]
    finally:
        q.destroy()


@pytest.mark.parametrize("with_queue", [True, False])
def test_recorded_ticks_keep_the_loop_unchanged(oracle, torch_dev, with_queue):
    """cvgs.attachQueueTicks(stream, queue, 16): the multi-camera loop as the reference's users write it -- one executeOperations(stream, ...)
    per camera, one fence per tick -- with 21 cameras on ONE stream: 16 calls go behind a gate when the 16th is recorded, the other 5 at the
    fence (launches: fewer than 8); the frames are rewritten on the stream between ticks; the consumers run on the stream behind the fence."""
    torch, dev = torch_dev
    cams = [Camera(torch, dev, oracle, seed=300 + k, n_crops=6, pool=3, wh=(640, 360)) for k in range(21)]
    stream = cams[0].stream
    for c in cams:
        c.stream = stream
        warm(torch, c)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        if with_queue:
            cvgs.attachQueueTicks(stream, q, 16)
        else:
            cvgs.recordTicks(stream, 16)  # no queue: one cvgs_execute_many launch per 16 recorded calls
        with torch.cuda.stream(stream):
            for i in range(40):
                for c in cams:
                    c.produce(i)
                for c in cams:
                    ops = H.k1_chain(cvgs.GpuMat.from_tensor(c.frame, cvgs.CV_8UC3), c.crops, cvgs.GpuMat.from_tensor(c.out, cvgs.CV_32FC1), DST, CN)
                    cvgs.executeOperations(stream, *ops)
                cvgs.fence(stream)
                for c in cams:
                    c.consume(i)
            for c in cams:  # recorded calls are submitted at detach as well
                c.produce(1)
            for c in cams:
                cvgs.executeOperations(stream, *H.k1_chain(cvgs.GpuMat.from_tensor(c.frame, cvgs.CV_8UC3), c.crops, cvgs.GpuMat.from_tensor(c.out, cvgs.CV_32FC1), DST, CN))
            cvgs.fence(stream)
            cvgs.detachQueue(stream)
            for c in cams:
                c.consume(1)
        stream.synchronize()
        st = q.stats()
        assert st["error"] == 0 and st["submitted"] == (41 * 16 if with_queue else 0), st
        assert sum(int(c.bad.item()) for c in cams) == 0
    finally:
        q.destroy()


def test_destroy_with_a_gate_kernel_still_behind_its_producer(oracle, torch_dev):
    """cvgs_queue_destroy while a stream-ordered batch's gate kernel has not run yet (a 150 ms producer in front of it): the destroy waits
    its bounded time, releases whatever is waiting and drains the device BEFORE it frees the words those kernels read -- no fault, the
    stream finishes, and a new queue serves the same stream afterwards."""
    torch, dev = torch_dev
    cam = Camera(torch, dev, oracle, seed=150)
    warm(torch, cam)
    lib = capi.load_library()
    q = cvgs.Queue(idle_us=5000.0)
    with torch.cuda.stream(cam.stream):
        H.aid_check(H.testaid().cvgs_debug_occupy(1, 64, 0, 150000.0, cam.stream.cuda_stream))
        cam.produce(1)
        q.submit_lowered_on(cam.stream, cam.lowered)
    t0 = time.perf_counter()
    q.destroy()
    assert time.perf_counter() - t0 < 8.0
    cam.stream.synchronize()
    torch.cuda.synchronize()
    q2 = cvgs.Queue(idle_us=5000.0)
    try:
        cam.bad.zero_()
        with torch.cuda.stream(cam.stream):
            for i in range(20):
                cam.produce(i)
                q2.submit_lowered_on(cam.stream, cam.lowered)
                cam.consume(i)
        cam.stream.synchronize()
        assert int(cam.bad.item()) == 0 and q2.stats()["error"] == 0
    finally:
        q2.destroy()


def test_a_tick_larger_than_the_ring(oracle, torch_dev):
    """16 chains behind one call on an 8-slot ring: the call splits them over several gate kernels (half a ring each) instead of waiting
    for its own first slot behind a gate nobody has launched."""
    torch, dev = torch_dev
    s = torch.cuda.Stream()
    cams = [Camera(torch, dev, oracle, seed=160 + k, n_crops=5, pool=2) for k in range(16)]
    for c in cams:
        c.stream = s
        warm(torch, c)
    ptrs = cvgs.Queue.chain_pointers([c.lowered for c in cams])
    q = cvgs.Queue(depth=8, idle_us=5000.0)
    try:
        with torch.cuda.stream(s):
            for i in range(12):
                for c in cams:
                    c.produce(i)
                q.submit_many_on(s, ptrs, len(cams))
                for c in cams:
                    c.consume(i)
        s.synchronize()
        assert q.stats()["error"] == 0 and q.stats()["submitted"] == 12 * 16
        assert sum(int(c.bad.item()) for c in cams) == 0
    finally:
        q.destroy()


def test_four_host_threads_submit_stream_ordered_batches(oracle, torch_dev):
    """Four host threads, each with its own camera stream, submit stream-ordered batches to ONE queue concurrently (a fifth submits
    ticket batches): ring order must stay equal to gate-kernel order under the races (the group lock), every tensor bit-exact."""
    import threading
    torch, dev = torch_dev
    cams = [Camera(torch, dev, oracle, seed=170 + k, n_crops=6 + 2 * k, pool=4) for k in range(4)]
    extra = Camera(torch, dev, oracle, seed=179, n_crops=9, pool=1)
    for c in cams + [extra]:
        warm(torch, c)
    extra.frame.copy_(extra.pool[0])
    torch.cuda.synchronize()
    q = cvgs.Queue(idle_us=5000.0)
    errors = []

    def camera_thread(c):
        try:
            with torch.cuda.stream(c.stream):
                for i in range(150):
                    c.produce(i)
                    q.submit_lowered_on(c.stream, c.lowered)
                    c.consume(i)
            c.stream.synchronize()
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    def ticket_thread():
        try:
            for i in range(100):
                q.wait(q.submit_lowered(extra.lowered), timeout_s=20.0)
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    try:
        ths = [threading.Thread(target=camera_thread, args=(c,)) for c in cams] + [threading.Thread(target=ticket_thread)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not errors, errors
        assert not any(t.is_alive() for t in ths)
        assert q.stats()["error"] == 0 and q.stats()["submitted"] == 4 * 150 + 100
        assert [int(c.bad.item()) for c in cams] == [0, 0, 0, 0]
        assert bool(torch.equal(extra.out.view(torch.int32), extra.refs[0]))
    finally:
        q.destroy()


def test_a_dependent_group_is_not_run_concurrently(oracle, torch_dev):
    """the server runs the batches of one group concurrently, so a group in which one chain writes what another chain writes (or
    reads) must not go behind ONE gate.  Two chains with the SAME output tensor: refused without the hybrid policy (CVGS_ERR_UNSUPPORTED,
    nothing queued); with it they are launched one by one, in order -- the tensor then holds the LAST chain's result, as two ordered
    cvgs_execute calls would leave it.  An independent group of the same shape still goes to the server."""
    torch, dev = torch_dev
    a, b = Camera(torch, dev, oracle, seed=71, n_crops=9), Camera(torch, dev, oracle, seed=72, n_crops=9)
    warm(torch, a)
    warm(torch, b)
    # chain `b2`: camera b's picture and crops INTO camera a's tensor
    ops = H.k1_chain(cvgs.GpuMat.from_tensor(b.frame, cvgs.CV_8UC3), b.crops, cvgs.GpuMat.from_tensor(a.out, cvgs.CV_32FC1), DST, CN)
    b2 = cvgs.lower(ops)
    dep = cvgs.Queue.chain_pointers([a.lowered, b2])
    ind = cvgs.Queue.chain_pointers([a.lowered, b.lowered])
    q = cvgs.Queue(idle_us=5000.0)
    s = torch.cuda.Stream()
    try:
        with torch.cuda.stream(s):
            a.frame.copy_(a.pool[1])
            b.frame.copy_(b.pool[2])
        s.synchronize()
        with pytest.raises(Exception):   # strict group on the server (explicit MIN_GROUP keeps it there): refused
            q.submit_many_on(s, dep, 2, cvgs.Queue.MIN_GROUP(2))
        assert q.stats()["submitted"] == 0
        t = q.submit_many_on(s, dep, 2, cvgs.Queue.HYBRID | cvgs.Queue.DEFER_WAIT | cvgs.Queue.MIN_GROUP(2))
        assert t == cvgs.Queue.TICKET_DIRECT and q.stats()["submitted"] == 0
        s.synchronize()
        assert torch.equal(a.out.view(torch.int32), b.refs[2])   # the second chain's bits: the launches ran in order
        t = q.submit_many_on(s, ind, 2, cvgs.Queue.MIN_GROUP(2))
        s.synchronize()
        assert t != cvgs.Queue.TICKET_DIRECT and q.stats()["submitted"] == 2 and q.stats()["error"] == 0
        assert torch.equal(a.out.view(torch.int32), a.refs[1]) and torch.equal(b.out.view(torch.int32), b.refs[2])
    finally:
        q.destroy()


def test_a_stream_at_the_servers_priority_is_not_taken_by_the_server(oracle, torch_dev):
    """the server's stream has the highest stream priority; a caller stream of that priority (torch.cuda.Stream(priority=-1)) may share
    its hardware queue, where the gate kernel would never start (10 s, error 3).  Such a stream is refused at once (hybrid: direct launch)."""
    torch, dev = torch_dev
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    if lo == hi:
        pytest.skip("one stream priority level only")
    cam = Camera(torch, dev, oracle, seed=81)
    warm(torch, cam)
    cam.stream = torch.cuda.Stream(priority=hi)
    q = cvgs.Queue(idle_us=5000.0)
    try:
        with torch.cuda.stream(cam.stream):
            cam.produce(1)
            t0 = time.perf_counter()
            with pytest.raises(Exception):
                q.submit_lowered_on(cam.stream, cam.lowered)
            assert time.perf_counter() - t0 < 1.0
            assert q.submit_lowered_on(cam.stream, cam.lowered, cvgs.Queue.HYBRID | cvgs.Queue.MIN_GROUP(1)) == cvgs.Queue.TICKET_DIRECT
            cam.consume(1)
        cam.stream.synchronize()
        assert int(cam.bad.item()) == 0 and q.stats()["submitted"] == 0 and q.stats()["error"] == 0
    finally:
        q.destroy()
