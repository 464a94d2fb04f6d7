"""bench_dist.py -- the N > 1 leg of bench.py: BASELINE cfg #5 under torch.distributed (one process per GPU, RCCL).

Every rank owns a stream of resident 6K frames and 64 crops per step (weak scaling: per-GPU work fixed); rank r's K1
launch writes rows [r*64, (r+1)*64) of the step's [N*64,3,128,64] tensor; the tensor must end up complete on EVERY
GPU.  Three legs, each K steps timed between barrier + torch.cuda.synchronize() on both sides, repeated, median taken,
max over ranks:
  compute_only : K1 alone, graph-replayed (what the sharded kernel scales to with no exchange at all)
  allgather    : K1 + in-place RCCL all-gather of the N slices per step (BASELINE's spelling; the collective runs on
                 RCCL's stream, so the K1 of later steps overlaps it)
  p2p_write    : K1 stores its rows into its own copy AND into every peer's copy through IPC-mapped pointers
                 (cvgs_write_desc.mirrors; SURVEY.md 8e option 2); "all rows of step k have landed" is a DEVICE-side flag
                 exchange over the same mappings (cvgs_exchange_signal / cvgs_exchange_wait: two tiny kernels on the launch
                 stream, no collective and no host round trip per step; round 2 used one RCCL all-reduce per step, 32 us)
  flags_only   : the flag exchange alone (no K1): what the protocol itself costs per step
`value` is the faster of the two legs that deliver the assembled tensor; every leg is reported.  --exchange-half assembles an
fp16 tensor (the half-precision hand-off): half the bytes on every xGMI link."""
import json
import os
import time

import numpy as np
import torch

import ctypes as C

import bench as B
from cvgpuspeedup_amd import capi, rccl, sharding
from cvgpuspeedup_amd import workloads as W


def _median_wall(fn, reps, dist, dev):
    """fn() enqueues K steps and drains its collectives; each rep sits between barrier + synchronize; max over ranks."""
    ts = []
    for _ in range(reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        ts.append(time.perf_counter() - t0)
    t = torch.tensor([float(np.median(ts))], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rotation_length(frames_arg, n_crops, esz, one_gpu=False):
    """(frames in rotation, read-touched bytes per frame, written bytes per frame) -- the SAME on every rank by construction: no argument
    depends on the rank (the per-rank crop lists would give lengths that differ by one or two)."""
    rd_frame, wr_frame = W.k1_touched_per_frame(n_crops, W.FRAME_6K, 0, esz)
    n_frames = frames_arg or max(6, W.rotation_units(rd_frame))
    if one_gpu and not frames_arg:
        n_frames = 8  # (test mode: N ranks share one GPU and gloo moves the tensors through the host -- the timings mean nothing, keep it short)
    return int(n_frames), rd_frame, wr_frame


def main(a, dev, rank, world):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import datetime
    # a rank that dies must not hold the others in a collective for the default 10 minutes
    # CVGS_BENCH_WORLD_ON_ONE_GPU=1 (tests/test_gpu_bench_world2.py): every rank on cuda:0 and gloo for the collectives (RCCL refuses two
    # ranks on one device) -- the IPC mappings, mirror stores, arrival flags, link probe and the line's assembly run as with N GPUs;
    # the timings of such a run mean nothing.
    one_gpu = os.environ.get("CVGS_BENCH_WORLD_ON_ONE_GPU") == "1"
    if one_gpu:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))
    else:
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
    n = B.CFG5_CROPS
    plane = 3 * W.DST[0] * W.DST[1]
    fw, fh = W.FRAME_6K
    half = bool(getattr(a, "exchange_half", False))
    esz = 2 if half else 4
    tensor_bytes = world * n * plane * esz
    per_frame_bytes = fw * fh * 3 + tensor_bytes
    # rotation sized from TOUCHED bytes (W.rotation_units: the read-touched set alone >= 2 x the Infinity Cache), as the N = 1 headline
    # (from RANK 0's crop lists on every rank: the rotation's length fixes the layout of the shared allocation -- tensors, then the flag
    #  words -- that the peers map, so every rank must arrive at the same number; per-rank crop lists gave 46 / 47 / 48 and a memory fault)
    n_frames, rd_frame, wr_frame = rotation_length(a.frames, n, esz, one_gpu)
    agree = [None] * world
    dist.all_gather_object(agree, int(n_frames))
    assert len(set(agree)) == 1, "ranks disagree on the rotation's length: %r" % (agree,)
    # how many ranks / distinct GPUs this run really spans (RCCL has only ever seen one rank on this pool)
    seen = [None] * world
    dist.all_gather_object(seen, (str(getattr(torch.cuda.get_device_properties(dev), "uuid", None) or dev), dist.get_backend()))
    rccl_ranks_seen = world if dist.get_backend() == "nccl" else 0
    gpus_seen = len({u for u, _ in seen})
    steps, reps = a.steps, 9

    # the step's full tensors live in ONE dedicated allocation per rank, so that peers can map it with one IPC handle
    flag_off = n_frames * tensor_bytes          # behind the tensors: one 8-byte arrival word per source rank, 128 bytes apart
    buf = rccl.DeviceBuffer(flag_off + world * 128 + 256)
    out_all = [buf.tensor(f * tensor_bytes, (world * n, plane), "<f2" if half else "<f4", esz) for f in range(n_frames)]
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = torch.cuda.current_stream().cuda_stream
    result_extra = {}

    # ---- leg 1: compute only (graph replay, the headline's own clock) --------------------------------------------
    wl = B.Workload(dev, n_frames, n, rank, world, False, frame_wh=W.FRAME_6K, out_all=out_all, half=half)
    m = B.measure(wl, steps, a.warmup, barrier=dist.barrier, target_s=0.1, min_replays=20)
    t = torch.tensor([m["step_s"]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    compute_step = float(t.item())
    px_step = n * W.DST[0] * W.DST[1] * world
    # ---- leg 1b: the same shard through the descriptor queue (the headline's kernel and submission path: one cvgs_queue_submit per step,
    # the rank's rows of the full tensor as the target); no barrier inside -- every rank times its own server, the MAX is taken after
    compute_queue_step = None
    try:
        dist.barrier()
        mq = B.measure_queue(wl, steps, a.warmup, target_s=0.1, min_replays=20, events=False)
        tq = torch.tensor([mq["step_s"] if B.queue_outputs_match_execute(wl) and not mq["queue"]["error"] else -1.0], dtype=torch.float64, device=dev)
    except Exception:
        tq = torch.tensor([-1.0], dtype=torch.float64, device=dev)
    tq_min = tq.clone()
    dist.all_reduce(tq, op=dist.ReduceOp.MAX)
    dist.all_reduce(tq_min, op=dist.ReduceOp.MIN)
    if float(tq_min.item()) > 0:
        compute_queue_step = float(tq.item())

    # ---- leg 2: K1 + in-place RCCL all-gather per step -----------------------------------------------------------------
    # The collective goes through the C-ABI (libcvgs_rccl.so: cvgs_comm_init_rank + cvgs_allgather_inplace = ncclAllGather in place), the
    # spelling a C++ host uses (examples/sharded_crops.cpp) -- torch.distributed only carries the 128-byte unique id to the
    # ranks.  It runs on a second stream behind an event of the step's K1, so the K1 of later steps overlaps it; a buffer's next K1 waits
    # for the collective that last read it.
    # (a communicator that cannot be created within 120 s -- on any rank -- leaves the leg to torch.distributed's own collective, and the line says
    #  which one ran: no N > 1 node was ever available to try this path on)
    native, allgather_via = None, "gloo (every rank on one GPU: test mode)"
    if not one_gpu:
        import threading
        box = {}
        try:
            uid = [rccl.Communicator.unique_id() if rank == 0 else None]
        except Exception as ex:
            uid, box["error"] = [None], repr(ex)
        dist.broadcast_object_list(uid, src=0)

        def make():
            try:
                torch.cuda.set_device(dev)
                box["comm"] = rccl.Communicator(world, rank, uid[0])
            except Exception as ex:
                box["error"] = repr(ex)
        if uid[0] is not None:
            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(120.0)
        okn = torch.tensor([1 if box.get("comm") is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okn, op=dist.ReduceOp.MIN)
        if int(okn.item()) == 1:
            native = box["comm"]
            rccl_ranks_seen = int(native.lib.cvgs_comm_size(native.handle))
            allgather_via = "libcvgs_rccl.so (cvgs_comm_init_rank + cvgs_allgather_inplace)"
        else:
            allgather_via = "torch.distributed all_gather_into_tensor (libcvgs_rccl.so's communicator was not created: %s)" % box.get("error", "timeout")
    coll = torch.cuda.Stream()
    k1_done = [torch.cuda.Event() for _ in range(n_frames)]
    gathered = [None] * n_frames
    slice_bytes = n * plane * esz

    def ag_steps():
        launch_stream = torch.cuda.current_stream()
        for i in range(steps):
            j = i % n_frames
            if gathered[j] is not None:
                launch_stream.wait_event(gathered[j])  # the collective that last read buffer j
            wl.launch(i, s)
            lo, hi = sharding.shard_bounds(world * n, world, rank)
            if one_gpu:  # gloo: no in-place CUDA all-gather -- through a copy, synchronously (test mode only)
                torch.cuda.synchronize()
                parts = [torch.empty_like(out_all[j][lo:hi]) for _ in range(world)]
                dist.all_gather(parts, out_all[j][lo:hi].clone())
                for r, part in enumerate(parts):
                    rlo, rhi = sharding.shard_bounds(world * n, world, r)
                    out_all[j][rlo:rhi].copy_(part)
                continue
            k1_done[j].record(launch_stream)
            coll.wait_event(k1_done[j])
            if native is not None:
                native.allgather_inplace(out_all[j].data_ptr(), slice_bytes, coll.cuda_stream)
            else:
                with torch.cuda.stream(coll):
                    dist.all_gather_into_tensor(out_all[j], out_all[j][lo:hi])
            if gathered[j] is None:
                gathered[j] = torch.cuda.Event()
            gathered[j].record(coll)
        coll.synchronize()

    ag_steps()
    ag_wall = _median_wall(ag_steps, reps, dist, dev)
    torch.cuda.synchronize()
    reference = out_all[0].clone()  # assembled by the collective: what the P2P leg must reproduce bit for bit

    # ---- leg 3: P2P fused write + one tiny all-reduce per step ---------------------------------------------------------
    p2p = {"ok": False}
    peers, bases, map_error = [], None, None
    handles = [None] * world
    dist.all_gather_object(handles, buf.handle())
    try:  # purely local: map every peer's allocation into this process
        bases = []
        for r in range(world):
            bases.append(buf.ptr if r == rank else rccl.open_peer(handles[r]))
            if r != rank:
                peers.append(bases[-1])
        # probe: a plain store into every peer's allocation (its last 64 bytes: nothing reads them) before any kernel is pointed at it --
        # with the engine's own copy kernel (a torch view of a peer's pointer would be created on the PEER's device and copied over)
        lib_p = capi.load_library()
        for b in peers:
            capi.check(lib_p.cvgs_stream_copy(b + flag_off + world * 128 + 128, buf.ptr + flag_off + world * 128 + 128, 64, s))
        torch.cuda.synchronize()
    except Exception as ex:  # no peer access on this box, IPC refused, ...
        map_error = repr(ex)
    mapped = torch.tensor([0 if map_error else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(mapped, op=dist.ReduceOp.MIN)  # every rank takes the same branch below: no collective is entered alone
    try:
        if int(mapped.item()) != 1:
            raise RuntimeError(map_error or "a peer could not map the tensors")
        # one xGMI link, measured (SURVEY.md 8e quotes ~153 GB/s per link, round 1 quoted 64 GB/s per direction -- neither had been
        # measured): every rank streams 64 MiB (or what the allocation holds) into its right-hand neighbour's allocation with the
        # engine's copy kernel, all ranks at once (each link carries one stream per direction), before any tensor is live
        if world > 1:
            local_gbs, nbytes = -1.0, 0
            try:  # purely local work; the collectives below are entered by EVERY rank whatever happened here
                nxt = bases[(rank + 1) % world]
                nbytes = min(64 << 20, (n_frames * tensor_bytes) & ~0xfff)
                lib0 = capi.load_library()
                for _ in range(2):
                    capi.check(lib0.cvgs_stream_copy(nxt, buf.ptr, nbytes, s))
                torch.cuda.synchronize()
            except Exception:
                nbytes = 0
            dist.barrier()
            try:
                if nbytes:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(8):
                        capi.check(lib0.cvgs_stream_copy(nxt, buf.ptr, nbytes, s))
                    e1.record()
                    torch.cuda.synchronize()
                    local_gbs = 8.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
            except Exception:
                local_gbs = -1.0
            lo_t = torch.tensor([local_gbs], dtype=torch.float64, device=dev)
            hi_t = lo_t.clone()
            dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
            if float(lo_t.item()) > 0:
                result_extra["xgmi_probe"] = {"GB_per_s_per_link_one_direction_min": round(float(lo_t.item()), 1), "max": round(float(hi_t.item()), 1),
                                              "bytes": int(nbytes), "pattern": "rank r -> rank r+1, all ranks at once, cvgs_stream_copy into the IPC mapping"}
            else:
                result_extra["xgmi_probe"] = {"error": "the copy into a peer's mapping failed on at least one rank"}
            dist.barrier()
        mirrors = [[bases[r] + f * tensor_bytes for r in range(world) if r != rank] for f in range(n_frames)]
        for o in out_all:
            o.zero_()
        torch.cuda.synchronize()
        dist.barrier()
        wl2 = B.Workload(dev, n_frames, n, rank, world, False, frame_wh=W.FRAME_6K, out_all=out_all, share=wl, mirrors=mirrors, half=half)
        lib = capi.load_library()
        others = [r for r in range(world) if r != rank]
        # my arrival word in every peer's flag block / the peers' words in mine
        sig = (C.c_void_p * max(1, len(others)))(*[bases[r] + flag_off + rank * 128 for r in others])
        own = (C.c_void_p * max(1, len(others)))(*[buf.ptr + flag_off + r * 128 for r in others])
        err_words = torch.zeros(2, dtype=torch.int64, device=dev)
        lag = max(1, min(4, n_frames - 1))  # waits trail the signals: a well-fed stream never blocks on them
        counter = torch.zeros(1, dtype=torch.int64, device=dev)  # the step number lives on the device: the K steps are ONE replayable graph

        def enqueue(with_k1):
            s = torch.cuda.current_stream().cuda_stream  # the capturing stream
            for i in range(steps):
                if with_k1:
                    wl2.launch(i, s)
                # my rows of this step have landed everywhere: advance the device-side step count, publish it to every peer
                capi.check(lib.cvgs_exchange_step(sig, own, len(others), counter.data_ptr(), lag, 2000.0, err_words.data_ptr(), s))
            capi.check(lib.cvgs_exchange_wait(own, len(others), 0, counter.data_ptr(), 0, 2000.0, err_words.data_ptr(), s))  # the last steps are complete here

        graphs = {}
        for with_k1 in (True, False):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue(with_k1)
            graphs[with_k1] = g

        def p2p_steps(with_k1=True):
            graphs[with_k1].replay()

        p2p_steps()
        torch.cuda.synchronize()
        dist.barrier()
        same = bool(torch.equal(out_all[0].view(torch.int32), reference.view(torch.int32)))
        okt = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        p2p_wall = _median_wall(p2p_steps, reps, dist, dev)
        flags_wall = _median_wall(lambda: p2p_steps(False), reps, dist, dev)
        torch.cuda.synchronize()
        lost = int(err_words[0].item())
        p2p = {"ok": bool(okt.item() == 1) and lost == 0, "wall": p2p_wall, "flags_wall": flags_wall, "kernel": wl2.kernel,
               "error": "a flag wait timed out (peer %d)" % int(err_words[1].item()) if lost else None}
    except Exception as ex:  # no peer access on this box, IPC refused, ...: the all-gather leg stands
        p2p = {"ok": False, "error": repr(ex)}
    okall = torch.tensor([1 if p2p.get("ok") else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(okall, op=dist.ReduceOp.MIN)
    p2p_ok = bool(okall.item() == 1)

    best_wall, exchange = ag_wall, "RCCL in-place all-gather per step via %s, on a second stream, overlapped with later steps' K1" % allgather_via
    if p2p_ok and p2p["wall"] < ag_wall:
        best_wall, exchange = p2p["wall"], "P2P fused write (K1 stores into every peer's tensor) + device-side arrival flags (no collective per step)"
    step_s = best_wall / steps
    result = None
    if rank == 0:
        alg = wl.algorithmic_bytes()
        link = (result_extra.get("xgmi_probe") or {}).get("GB_per_s_per_link_one_direction_min")
        result = {
            "metric": B.baseline_metric(), "value": round(px_step / step_s / 1e6, 1), "unit": "Mpix/s", "n_gpus": world,
            "steps": steps, "warmup": a.warmup, "ms_per_step": round(step_s * 1e3, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",  # (arithmetic in fp32; --exchange-half only narrows the stored tensor)
            "config": {"workload": "cfg5: %d x (64 variable-size crops of a resident 6K u8c3 frame per GPU) -> [%d,3,128,64] fp32 "
                                   "assembled on every GPU per step; %d resident frames per GPU" % (world, world * n, n_frames) + (" (fp16 tensor: --exchange-half)" if half else ""),
                       "chain": "resize(bilinear) -> RGB2BGR -> x0.3 -> -(1,4,3.2) -> /(3.2,0.6,11.8) -> TensorSplit",
                       "crops_per_launch": n, "frame": "6144x3456 u8c3", "kernel": wl.kernel, "exchange": exchange,
                       "parallelism": "1 process per GPU, crop lists sharded, frames never replicated"},
            "roofline": {"bound": "hbm", "achieved": round(alg / compute_step / 1e9, 1), "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / compute_step / 1e9 / B.HBM_PEAK_GBS, 4), "traffic": None, "kernel": wl.kernel,
                         "kernel_us": round(compute_step * 1e6, 3), "algorithmic_bytes_per_launch": int(alg),
                         "per_gpu_frac": round(alg / compute_step / 1e9 / B.HBM_PEAK_GBS, 4),
                         "per_gpu_frac_on_the_queue": round(alg / compute_queue_step / 1e9 / B.HBM_PEAK_GBS, 4) if compute_queue_step else None,
                         "residency": W.residency(n_frames, rd_frame, wr_frame, per_frame_bytes),
                         "note": "K1 alone on each GPU (compute-only leg); the exchange is xGMI-bound, not HBM-bound"},
            # the SAME workload and submission path on ONE GPU (this rank's 64-crop-of-6K step, graph-replayed launches, measured in this
            # process while the other ranks run theirs): what a scaling curve of this line must be read against -- bench.py --gpus 1 is
            # cfg #2b on the descriptor queue, a different workload and submission path
            "n1_same_workload": {"Mpix_per_s": round(n * W.DST[0] * W.DST[1] / compute_step / 1e6, 1), "us_per_step": round(compute_step * 1e6, 3),
                                 "on_the_queue_Mpix_per_s": round(n * W.DST[0] * W.DST[1] / compute_queue_step / 1e6, 1) if compute_queue_step else None,
                                 "on_the_queue_us_per_step": round(compute_queue_step * 1e6, 3) if compute_queue_step else None,
                                 "value_over_n1": round((px_step / step_s) / (n * W.DST[0] * W.DST[1] / compute_step), 3)},
            "rccl_ranks_seen": rccl_ranks_seen, "gpus_seen": gpus_seen,
            "legs": {"compute_only_us": round(compute_step * 1e6, 3), "compute_only_on_the_queue_us": round(compute_queue_step * 1e6, 3) if compute_queue_step else None,
                     "allgather_us": round(ag_wall / steps * 1e6, 3),
                     "p2p_write_us": round(p2p["wall"] / steps * 1e6, 3) if p2p_ok else None,
                     "link_floor_us": round(n * plane * esz / (link * 1e9) * 1e6, 3) if link else None},
            "extra": {
                "compute_only": {"Mpix_per_s": round(px_step / compute_step / 1e6, 1), "us_per_step": round(compute_step * 1e6, 3),
                                 "note": "graph-replayed K1, no exchange: every rank keeps its shard"},
                "allgather": {"Mpix_per_s": round(px_step * steps / ag_wall / 1e6, 1), "us_per_step": round(ag_wall / steps * 1e6, 3), "via": allgather_via,
                              "bytes_received_per_gpu_per_step": (world - 1) * n * plane * esz},
                "p2p_write": ({"Mpix_per_s": round(px_step * steps / p2p["wall"] / 1e6, 1), "us_per_step": round(p2p["wall"] / steps * 1e6, 3),
                               "matches_allgather_bit_exact": True, "kernel": p2p.get("kernel"),
                               "barrier": "device-side arrival flags over the IPC mappings (cvgs_exchange_step: ONE one-wave kernel per step, none with a single rank), waits trail by <= 4 steps",
                               "flags_only_us_per_step": round(p2p["flags_wall"] / steps * 1e6, 3)} if p2p_ok else
                              {"error": p2p.get("error", "result differs from the all-gather's or a rank failed")}),
                # every GPU receives world-1 slices, each from a different peer over its own xGMI link (~153 GB/s)
                "xgmi_floor_us_per_step": round(n * plane * esz / 153e9 * 1e6, 3) if world > 1 else 0.0,
                "tensor_element_bytes": esz,
            }}
        result["extra"].update(result_extra)
        if "xgmi_probe" in result_extra:
            result["xgmi_probe"] = result_extra["xgmi_probe"]
    dist.barrier()
    if native is not None:
        native.destroy()
    for p in peers:
        try:
            rccl.close_peer(p)
        except Exception:
            pass
    dist.destroy_process_group()
    # the line is the LAST thing rank 0 writes: after the process group is gone (RCCL's banner and teardown messages are behind us;
    # bench.guard_stdout has routed every other write of every rank to stderr)
    if result is not None:
        B.emit(result, a)
