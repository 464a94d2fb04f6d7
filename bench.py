#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json, measured on MI355X.

Metric: Mpixels/s of the fused crop+resize+normalize+split kernel (K1), 50 variable-size crops of a 4K
frame -> [50,3,128,64] fp32 NCHW (BASELINE.md cfg #2b), plus the fraction of the HBM roofline the kernel
reaches and the CPU restatement timed beside it.

A "step" = one pass of the hot path over one batch = ONE cvgs_execute() = one K1 launch over the 50 crops
of one frame.  Steps cycle over enough distinct resident frames / outputs to exceed 2x the 256 MB Infinity
Cache, so reads and writes really go to HBM.  Inputs (frames, crop descriptors) are resident in HBM before
the timed region; the K timed steps are replayed from HIP graphs so the host's launch rate is not what is
measured (eager numbers are reported next to it under "extra").

N > 1 (one process per GPU, torch.distributed/RCCL): each rank owns its own frames and crop lists (weak
scaling), runs K1 into its slice of the [N*50,3,128,64] tensor and all-gathers the slices over xGMI every
step (BASELINE.json north_star).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs, sharding  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
CROPS = 50
INFINITY_CACHE = 256 << 20


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo); an ASCII spelling if it is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mpixels/s fused crop+resize+norm+split, 50x->64x128 NCHW; % HBM roofline @1/2/4/8 GPU"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4096)
    p.add_argument("--warmup", type=int, default=256)
    p.add_argument("--crops", type=int, default=CROPS, help="crops per launch (headline: 50)")
    p.add_argument("--frames", type=int, default=0, help="distinct resident frames (0 = enough to defeat the cache)")
    p.add_argument("--table", action="store_true", help="descriptors in a resident device table, not kernel args")
    p.add_argument("--eager", action="store_true", help="time eager launches instead of graph replay")
    p.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    p.add_argument("--no-extra", action="store_true", help="skip the extra sweeps")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--force-dist", action="store_true", help="take the torch.distributed path even with one rank (testing)")
    return p.parse_args()


class Workload:
    """F resident 4K frames, F crop lists, F output tensors, F pre-lowered chains."""

    def __init__(self, dev, n_frames, crops_per_launch, rank, world, use_table, frame_wh=W.FRAME_4K,
                 out_all=None, flags=0, share=None, half=False):
        self.dev = dev
        fw, fh = frame_wh
        self.frames, self.outs, self.chains, self.crops, self.tables = [], [], [], [], []
        self.lib = capi.load_library()
        self.n = crops_per_launch
        plane = 3 * W.DST[0] * W.DST[1]
        for f in range(n_frames):
            seed = W.SEED + 1000 * rank + f
            frame = share.frames[f] if share is not None else W.random_u8_torch((fh, fw, 3), seed, dev)
            crops = W.random_crops(crops_per_launch, fw, fh, seed=seed + 500000)
            if out_all is not None:  # in-place all-gather layout: this rank's slice of the full tensor
                out = out_all[f][rank * crops_per_launch:(rank + 1) * crops_per_launch]
            else:
                out = torch.zeros((crops_per_launch, plane), dtype=torch.float16 if half else torch.float32, device=dev)
            g_src = cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3)
            g_out = cvgs.GpuMat.from_tensor(out, cvgs.CV_16FC1 if half else cvgs.CV_32FC1)
            ops = W.k1_chain(g_src, crops, g_out, half=half)
            if use_table:
                tab = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(dev)
                self.tables.append(tab)
                ops = W.k1_chain(g_src, crops, g_out, table=tab.data_ptr(), half=half)
            self.frames.append(frame)
            self.outs.append(out)
            self.crops.append(crops)
            self.chains.append(cvgs.lower(ops, flags))
        self.kernel = cvgs.kernel_name(*ops, flags=flags)

    def launch(self, i, stream):
        ch = self.chains[i % len(self.chains)]
        rc = self.lib.cvgs_execute(C.byref(ch.desc), stream)
        if rc:
            capi.check(rc)


def make_graphs(wl, steps, chunk=256):
    """Capture the K steps as HIP graphs (chunks of <= 256 launches): returns [(graph, n_launches, repeats)]."""
    plan = []
    full, rem = divmod(steps, chunk)
    base = 0
    for n, reps in ((chunk, full), (rem, 1)):
        if n == 0 or reps == 0:
            continue
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s = torch.cuda.current_stream().cuda_stream
            for i in range(n):
                wl.launch(base + i, s)
        plan.append((g, n, reps))
        base += n
    return plan


def run_steps(wl, steps, eager, plan=None):
    if eager:
        s = torch.cuda.current_stream().cuda_stream
        for i in range(steps):
            wl.launch(i, s)
    else:
        for g, _, reps in plan:
            for _ in range(reps):
                g.replay()


def timed(fn, dist_barrier):
    """barrier + synchronize on both sides; returns (wall seconds, device seconds by HIP events)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist_barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    dist_barrier()
    t1 = time.perf_counter()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


def cpu_quota():
    """CPUs this container may actually use (cgroup v2 cpu.max), or None if unlimited / unreadable: the thread count of
    the CPU leg is the host's, but a quota caps what those threads can deliver."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else round(float(quota) / float(period), 2)
    except Exception:
        return None


def cpu_baseline(wl, seconds):
    """The CPU restatement (oracle, kind 'port') timed on this box's host cores on a bounded sample of the same
    workload: frame 0's 50-crop batch, repeated for ~`seconds`.  Timed code = oracle_k1_fast_repeat, the headline chain
    as a plain C loop nest (OpenMP over passes x crops x rows); the descriptor interpreter oracle_execute is the checker:
    it must agree with the loop nest bit for bit, and the GPU output of that batch must agree with both."""
    from oracle import oracle_binding as ob
    lib = ob.load_oracle()
    cores = lib.oracle_max_threads()
    quota = cpu_quota()
    if quota:  # more runnable threads than the container's CPU quota only get throttled
        cores = max(1, min(cores, int(quota + 0.999)))
    lib.oracle_set_threads(cores)
    frame = wl.frames[0].cpu().numpy()
    ref = np.zeros((wl.n, 3 * W.DST[0] * W.DST[1]), np.float32)
    fast = np.zeros_like(ref)

    def lowered(out):
        return cvgs.lower(W.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), wl.crops[0], cvgs.GpuMat.from_array(out, cvgs.CV_32FC1)))

    chain, chain_fast = lowered(ref), lowered(fast)
    ob.execute(chain)                 # the checker (interpreter)
    ob.execute_k1_fast(chain_fast)    # the timed implementation, warm
    agree = bool((ref.view(np.uint32) == fast.view(np.uint32)).all())
    per_call = max(8, 2 * cores)      # passes per parallel region
    reps, t0 = 0, time.perf_counter()
    while True:
        ob.execute_k1_fast(chain_fast, per_call)
        reps += per_call
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 10_000_000:
            break
    px = wl.n * W.DST[0] * W.DST[1] * reps
    # the same batch on ONE host thread (SURVEY.md 8d asks for both): ~1/4 of the time budget
    lib.oracle_set_threads(1)
    reps1, t1 = 0, time.perf_counter()
    while True:
        ob.execute_k1_fast(chain_fast)
        reps1 += 1
        dt1 = time.perf_counter() - t1
        if dt1 >= seconds / 4 or reps1 >= 100000:
            break
    lib.oracle_set_threads(cores)
    single = wl.n * W.DST[0] * W.DST[1] * reps1 / dt1 / 1e6
    s = torch.cuda.current_stream().cuda_stream
    wl.launch(0, s)
    torch.cuda.synchronize()
    gpu = wl.outs[0].cpu().numpy()
    checked = bool((gpu.view(np.uint32) == ref.view(np.uint32)).all())
    return {"value": round(px / dt / 1e6, 2), "unit": "Mpix/s", "cores": int(cores), "kind": "port",
            "sample": "%d x the 50-crop batch of frame 0 (oracle_k1_fast_repeat: plain C loop nest, OpenMP %d threads, %.1f s)" % (
                reps, cores, dt),
            "single_thread_value": round(single, 2), "host_threads": int(lib.oracle_max_threads()), "cgroup_cpu_quota": quota, "loop_nest_matches_interpreter_bit_exact": agree,
            "gpu_matches_oracle_bit_exact": checked}


def algorithmic_bytes(wl, out_elem=4):
    """SURVEY.md 8d figure per launch (tap census in cvgpuspeedup_amd/workloads.py; tests cross-check it with the oracle's)."""
    per_launch = [W.k1_algorithmic_bytes(c, out_elem=out_elem) for c in wl.crops]
    return float(np.mean(per_launch))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    n = a.crops
    plane = 3 * W.DST[0] * W.DST[1]
    per_frame_bytes = W.FRAME_4K[0] * W.FRAME_4K[1] * 3 + n * plane * 4 * world
    n_frames = a.frames or max(8, (2 * INFINITY_CACHE + per_frame_bytes - 1) // per_frame_bytes + 1)

    out_all = None
    if use_dist:
        # the full [world*n, C*H*W] tensor of every in-flight step; rank r's K1 writes rows [r*n, (r+1)*n)
        out_all = [torch.zeros((world * n, plane), dtype=torch.float32, device=dev) for _ in range(n_frames)]
    wl = Workload(dev, n_frames, n, rank, world, a.table, out_all=out_all)

    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)

    if not use_dist:
        plan = None if a.eager else make_graphs(wl, a.steps)
        warm = None if a.eager else (make_graphs(wl, a.warmup) if a.warmup else [])
        run_steps(wl, a.warmup, a.eager, warm)
        if not a.eager:
            # instantiate + upload the timed graphs before the clock starts (one untimed replay): the first replay of a
            # HIP graph pays its one-time setup, which is not part of a step
            run_steps(wl, a.steps, False, plan)
        wall, dev_s = timed(lambda: run_steps(wl, a.steps, a.eager, plan), barrier)
        gather_note = None
    else:
        # Per step: K1 into this rank's slice of the step's tensor, then the in-place all-gather (RCCL over xGMI)
        # that assembles it on every rank.  The collective runs asynchronously on RCCL's stream, so the K1 of the
        # following steps overlaps it; a buffer is only rewritten after the gather that last used it has completed.
        s = torch.cuda.current_stream().cuda_stream
        works = [None] * n_frames

        def step(i):
            j = i % n_frames
            if works[j] is not None:
                works[j].wait()  # current stream waits for the collective that read buffer j
            wl.launch(i, s)
            works[j] = sharding_gather(out_all[j], world * n, dist)

        def sharding_gather(full, items, d):
            lo, hi = sharding.shard_bounds(items, world, rank)
            return d.all_gather_into_tensor(full, full[lo:hi], async_op=True)

        def drain():
            for w in works:
                if w is not None:
                    w.wait()

        for i in range(a.warmup):
            step(i)
        drain()
        wall, dev_s = timed(lambda: ([step(i) for i in range(a.steps)], drain()), barrier)
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        gather_note = "in-place all_gather_into_tensor of %d x %d B per step, overlapped with the next steps' K1" % (
            world, n * plane * 4)
        # the same K steps WITHOUT the collective (every rank keeps its shard): what the sharded K1 alone scales to.
        # Reported under "extra" only; `value` above includes the all-gather the north star asks for.
        s2 = torch.cuda.current_stream().cuda_stream
        for i in range(min(a.warmup, 64)):
            wl.launch(i, s2)
        wall_c, _ = timed(lambda: [wl.launch(i, s2) for i in range(a.steps)], barrier)
        tc = torch.tensor([wall_c], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        compute_only = {"value": round(n * W.DST[0] * W.DST[1] * world * a.steps / float(tc.item()) / 1e6, 1), "unit": "Mpix/s",
                        "note": "same K steps, K1 only, no all-gather (eager launches; each rank keeps its shard)"}

    px_per_step = n * W.DST[0] * W.DST[1] * world
    value = px_per_step * a.steps / wall / 1e6
    result = {
        "metric": baseline_metric(),
        "value": round(value, 1),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 6),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg2b: %d variable-size crops (w~U[32,512], h~U[64,1024]) of a 4K u8c3 frame -> "
                               "[%d,3,128,64] fp32 per launch; %d resident frames cycled (working set %.0f MB)" % (
                                   n, n, n_frames, n_frames * per_frame_bytes / 1e6),
                   "chain": "resize(bilinear) -> RGB2BGR -> x0.3 -> -(1,4,3.2) -> /(3.2,0.6,11.8) -> TensorSplit",
                   "crops_per_launch": n, "frame": "3840x2160 u8c3", "kernel": wl.kernel,
                   "descriptors": "device table" if a.table else "kernel arguments",
                   "submission": ("eager, one K1 launch + one all-gather per step" if use_dist else
                                  ("eager" if a.eager else "hipGraph replay (256-launch graphs)")),
                   "parallelism": "1 process per GPU, crop lists sharded, %s" % (gather_note or "no collective")},
    }

    if rank == 0:
        alg = algorithmic_bytes(wl)
        if not use_dist:
            k_s = dev_s / a.steps  # HIP events on the launch stream over the timed region / K launches
        else:
            # K1 alone, serialized on the launch stream (the timed region interleaves the gathers): 256 launches
            # between one pair of HIP events, same method as the single-GPU leg
            s = torch.cuda.current_stream().cuda_stream
            for i in range(32):
                wl.launch(i, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(256):
                wl.launch(i, s)
            e1.record()
            torch.cuda.synchronize()
            k_s = e0.elapsed_time(e1) * 1e-3 / 256
        achieved = alg / k_s / 1e9
        ceiling = copy_ceiling(dev)
        result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(n, a.table),
                              "kernel": wl.kernel, "kernel_us": round(k_s * 1e6, 3),
                              "algorithmic_bytes_per_launch": int(alg),
                              # the on-box device-to-device copy ceiling (read + write bytes / time), measured now
                              "copy_ceiling": ceiling, "frac_of_copy_ceiling": round(achieved / ceiling, 4) if ceiling else None}
    if use_dist:
        barrier()
        if rank == 0:
            result.setdefault("extra", {})["without_allgather"] = compute_only
            result["extra"]["per_step_gather_bytes_received_per_gpu"] = (world - 1) * n * plane * 4
    if rank == 0 and world == 1 and not a.no_cpu:
        result["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
    if rank == 0 and world == 1 and not a.no_extra:
        result.setdefault("extra", {}).update(extra_sweeps(dev, a))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def timing_distribution(wl, singles=200, bursts=100, burst=64):
    """SURVEY.md 8d timing protocol: hipEvent pairs around (i) single launches and (ii) bursts of 64 back-to-back
    launches cycling over the resident frames; median / p10 / p90 in microseconds per launch.  Eager submission, so
    the single-launch figures include the event records and the host's launch path; plus the host's enqueue time per
    cvgs_execute call (the quantity the reference's 'CPU' benchmark measures, benchmarks/benchmark_CPU_OpenCV_vs_cvGS.cu)."""
    s = torch.cuda.current_stream().cuda_stream

    def measure(n_launch, reps, base):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for r, (e0, e1) in enumerate(ev):
            e0.record()
            for i in range(n_launch):
                wl.launch(base + r * n_launch + i, s)
            e1.record()
        torch.cuda.synchronize()
        t = np.sort(np.array([e0.elapsed_time(e1) * 1e3 / n_launch for e0, e1 in ev]))
        return {"median": round(float(np.median(t)), 3), "p10": round(float(t[len(t) // 10]), 3),
                "p90": round(float(t[(len(t) * 9) // 10]), 3)}

    run_steps(wl, 64, True)
    torch.cuda.synchronize()
    out = {"single_launch_us": measure(1, singles, 0), "burst64_us_per_launch": measure(burst, bursts, 7)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2048):
        wl.launch(i, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out["host_enqueue_us_per_call"] = round((t1 - t0) / 2048 * 1e6, 3)
    return out


def copy_ceiling(dev, mib=256, iters=20):
    """GB/s (read + write bytes / time) of a plain device-to-device copy of `mib` MiB (working set 2x the Infinity Cache)
    on this box -- the better of the
    runtime's copy (torch copy_ = hipMemcpy DtoD) and the engine's streaming copy kernel (cvgs_stream_copy): what a
    perfectly streaming kernel gets out of this HBM, next to the 8 TB/s spec the roofline fraction is quoted on."""
    try:
        n = mib << 20
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        b = torch.empty(n, dtype=torch.uint8, device=dev)
        a.zero_()
        lib = capi.load_library()
        s = torch.cuda.current_stream().cuda_stream

        def own():
            capi.check(lib.cvgs_stream_copy(b.data_ptr(), a.data_ptr(), n, s))

        best = 0.0
        for fn in (lambda: b.copy_(a), own):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = max(best, 2.0 * n * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        torch.cuda.empty_cache()
        return round(best, 1)
    except Exception:
        return None


def pmc_traffic(crops, table):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE,
    separate --pmc runs, corrected as calibrated in profiles/r01_b_pmc_hbm.txt).  Counters cannot be read from inside
    this process, so the value is the one measured for exactly this workload/kernel; null for any other configuration."""
    path = os.path.join(ROOT, "profiles", "pmc_headline.json")
    if crops != CROPS or table or not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
        return int(j["fetch_size_kb"] * 1024 * j["fetch_correction"] + j["write_size_kb"] * 1024)
    except Exception:
        return None


def multi_stream(dev, n_streams=4, per_stream=64, rounds=16):
    """Independent batches submitted on several streams (one HIP graph with parallel branches): consecutive launches
    of ONE stream are serialised by the queue's barrier bit, so a single stream pays the full launch/drain latency per
    batch; independent streams overlap it.  Throughput only -- per-kernel durations stretch when kernels overlap."""
    wl = Workload(dev, 24, CROPS, 0, 1, use_table=False)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for i in range(64):
        wl.launch(i, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        for k, st in enumerate(streams):
            st.wait_stream(cap)
            with torch.cuda.stream(st):
                for i in range(per_stream):
                    wl.launch(k * per_stream + i, st.cuda_stream)
        for st in streams:
            cap.wait_stream(st)
    g.replay()
    wall, dev_s = timed(lambda: [g.replay() for _ in range(rounds)], lambda: None)
    launches = n_streams * per_stream * rounds
    return {"streams": n_streams, "us_per_batch": round(dev_s / launches * 1e6, 3),
            "Mpix_per_s": round(CROPS * 8192 * launches / wall / 1e6, 1)}


def extra_sweeps(dev, a):
    """Secondary measurements (not the headline): eager submission and larger crop lists per launch, where the
    kernel leaves the launch-latency regime (SURVEY.md 'hard parts': cfg #2 moves only ~6-10 MB per launch)."""
    out = {}
    try:
        # the 50-crop batch on 1080p and 6K source frames (the headline uses 4K); crop sizes are clipped to the frame
        for name, wh in (("frame_1080p_50", W.FRAME_1080P), ("frame_6k_50", W.FRAME_6K)):
            per_frame = wh[0] * wh[1] * 3 + 50 * 3 * 64 * 128 * 4
            nf = max(4, min(48, (2 * INFINITY_CACHE) // per_frame + 1))
            wl = Workload(dev, nf, 50, 0, 1, use_table=False, frame_wh=wh)
            plan = make_graphs(wl, 2048)
            run_steps(wl, 64, True)
            wall, dev_s = timed(lambda: run_steps(wl, 2048, False, plan), lambda: None)
            alg = algorithmic_bytes(wl)
            out[name] = {"Mpix_per_s": round(50 * 8192 * 2048 / wall / 1e6, 1), "kernel_us": round(dev_s / 2048 * 1e6, 3),
                         "GB_per_s": round(alg / (dev_s / 2048) / 1e9, 1), "frac": round(alg / (dev_s / 2048) / 1e9 / HBM_PEAK_GBS, 4)}
            del wl, plan
            torch.cuda.empty_cache()
        for crops in (50, 200, 800, 3200):
            per_frame = W.FRAME_4K[0] * W.FRAME_4K[1] * 3 + crops * 3 * 64 * 128 * 4
            nf = max(4, min(24, (2 * INFINITY_CACHE) // per_frame + 1))
            wl = Workload(dev, nf, crops, 0, 1, use_table=True)
            steps = max(32, 4096 * 50 // crops)
            steps -= steps % 1
            plan = make_graphs(wl, steps)
            run_steps(wl, min(steps, 64), True)
            wall, dev_s = timed(lambda: run_steps(wl, steps, False, plan), lambda: None)
            alg = algorithmic_bytes(wl)
            out["crops_per_launch_%d" % crops] = {
                "Mpix_per_s": round(crops * 8192 * steps / wall / 1e6, 1), "kernel_us": round(dev_s / steps * 1e6, 3),
                "GB_per_s": round(alg / (dev_s / steps) / 1e9, 1), "frac": round(alg / (dev_s / steps) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": wl.kernel}
            del wl, plan
            torch.cuda.empty_cache()
        # half-precision hand-off option: same chain + convertTo<CV_32FC3, CV_16FC3>, fp16 NCHW tensor
        for crops in (50, 3200):
            per_frame = W.FRAME_4K[0] * W.FRAME_4K[1] * 3 + crops * 3 * 64 * 128 * 2
            nf = max(4, min(24, (2 * INFINITY_CACHE) // per_frame + 1))
            wl = Workload(dev, nf, crops, 0, 1, use_table=crops > 64, half=True)
            steps = max(32, 4096 * 50 // crops)
            plan = make_graphs(wl, steps)
            run_steps(wl, min(steps, 64), True)
            wall, dev_s = timed(lambda: run_steps(wl, steps, False, plan), lambda: None)
            alg = algorithmic_bytes(wl, out_elem=2)
            out["fp16_output_%d" % crops] = {
                "Mpix_per_s": round(crops * 8192 * steps / wall / 1e6, 1), "kernel_us": round(dev_s / steps * 1e6, 3),
                "GB_per_s": round(alg / (dev_s / steps) / 1e9, 1), "frac": round(alg / (dev_s / steps) / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": wl.kernel}
            del wl, plan
            torch.cuda.empty_cache()
        wl = Workload(dev, 24, 50, 0, 1, use_table=False)
        run_steps(wl, 256, True)
        wall, dev_s = timed(lambda: run_steps(wl, 2048, True), lambda: None)
        out["eager_50"] = {"Mpix_per_s": round(50 * 8192 * 2048 / wall / 1e6, 1), "us_per_step": round(wall / 2048 * 1e6, 3),
                           "note": "python ctypes + cvgs_execute + hipLaunchKernel per step (host-bound)"}
        out["timing_distribution_50"] = timing_distribution(wl)
        del wl
        torch.cuda.empty_cache()
        out["multi_stream_50"] = multi_stream(dev)
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_more
        out["other_configs"] = bench_more.run_all(dev, iters=50)
        import bench_resize  # whole-frame resizes of synthetic 1080p / 4K / 6K frames (K2 / K3 chains of the reference's tests)
        out["whole_frame_resize"] = bench_resize.run_all(dev, iters=50)
    except Exception as ex:  # extras must never break the headline line
        out["error"] = repr(ex)
    return out


if __name__ == "__main__":
    main()
