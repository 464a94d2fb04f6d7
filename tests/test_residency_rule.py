"""The residency rule of round 5 (VERDICT r4 "What's weak" #2; SURVEY.md 8d: "working sets that defeat the 256 MB Infinity Cache"): benchmark
rotations are sized from the bytes a launch TOUCHES -- the distinct 64-byte sectors that hold a tapped byte -- not from whole frames; and the
pieces of bench.py's line that carry it.  Host-only."""
import json

import numpy as np

import bench
from cvgpuspeedup_amd import workloads as W


def test_rotation_units_rule():
    # 7.95 MB of tapped sectors per frame -> 68 frames for 2 x 256 MiB; round 4's 20 frames were 0.3 of that
    assert W.rotation_units(7_947_888) == 68
    assert W.rotation_units(7_947_888, requested=20) == 20
    assert W.rotation_units(10 ** 12) == 8 and W.rotation_units(10 ** 12, minimum=3) == 3
    assert W.rotation_units(1) == 2 * W.LLC_BYTES
    for rd in (1_000_000, 5_379_520, 15_040_512):
        assert W.rotation_units(rd) * rd >= 2 * W.LLC_BYTES > (W.rotation_units(rd) - 1) * rd


def test_headline_frames_touch_a_third_of_their_bytes():
    rd, wr = W.k1_touched_per_frame(50, W.FRAME_4K)
    assert wr == 50 * 3 * 4 * 64 * 128
    tapped = np.mean([sum(W.tapped_bytes(w, h, 64, 128, 3) for (_, _, w, h) in W.random_crops(50, 3840, 2160, seed=W.SEED + f + 500000)) for f in range(8)])
    assert tapped <= rd <= 3840 * 2160 * 3          # at least the tapped bytes, at most the frame
    assert 7.0e6 < rd < 9.0e6                        # ~7.95 MB: a third of the 24.9 MB frame
    r = W.residency(20, rd, wr, 3840 * 2160 * 3 + wr)
    assert r["touched_MB"] < r["llc_MB"] < r["whole_frames_MB"]   # round 4's rotation: 596 MB of frames, 257 MB touched, a 268 MB cache
    r96 = W.residency(96, rd, wr)
    assert r96["read_touched_MB"] > 2 * r96["llc_MB"]


def test_nv12_sector_census_is_consistent():
    w, h = 640, 360
    whole = W.nv12_sector_read_bytes(w, h, 64, 128)
    crops = W.nv12_crops_sector_read_bytes([(0, 0, w, h)], w, h)
    assert crops == whole                            # one crop = the whole surface: the two censuses agree
    two = W.nv12_crops_sector_read_bytes([(0, 0, w, h), (0, 0, w, h)], w, h)
    assert two == whole                              # overlapping crops share their sectors
    assert whole <= (h + h // 2) * w and whole % 64 == 0
    small = W.nv12_crops_sector_read_bytes([(64, 32, 128, 64)], w, h)
    assert 0 < small < whole


def test_tick_passes_cover_whole_ticks():
    assert bench.tick_passes(20, 16) == (16, 20)     # the driver's --steps 20: 320 steps = 20 launches of 16
    assert bench.tick_passes(256, 16) == (1, 16)
    assert bench.tick_passes(4096, 16) == (1, 256)
    for k in (1, 7, 20, 33, 100, 256, 1000):
        m, launches = bench.tick_passes(k, 16)
        assert (m * k) % 16 == 0 and m * k >= 256 and launches * 16 == m * k


def test_configs_block_and_compact_line():
    extra = {"other_configs": [
        {"config": "cfg4 CircularTensor depth 16, 1080p fp32 x3, push 1080p convert+normalize", "us_per_update": 120.0, "frac_of_8TBs": 0.83, "GB_per_s": 6666.0, "copy_same_footprint_GB_per_s": 6900.0},
        {"config": "cfg4 CircularTensor depth 16, 1080p fp32 x3 MIRRORED ring (opt-in, data() moves), push 1080p convert+normalize", "us_per_update": 15.0},
        {"config": "cfg3 NV12 6144x3456 -> BGR float -> 1280x720 -> normalize -> split, one kernel per frame (graph-replayed launches)", "us_per_launch": 7.9, "frac_of_8TBs": 0.35, "surfaces_in_rotation": 36, "output_Mpix_per_s": 116000.0},
        {"config": "cfg3 NV12 6144x3456 -> BGR float -> 1280x720 -> normalize -> split, a TICK of 4 cameras' surfaces per launch (one chain, batch 4; graph-replayed), time per frame", "us_per_launch": 5.9, "frac_of_8TBs": 0.47},
        {"config": "cfg3 NV12 6144x3456 -> BGR float -> 1280x720 -> normalize -> split, one cvgs_queue_submit per frame (descriptor queue, no launch per frame)", "us_per_launch": 4.9, "frac_of_8TBs": 0.56},
        {"config": "cfg3 P010 (10-bit, BT.2020 limited) 6144x3456 -> ...", "us_per_launch": 10.5}]}
    c = bench.configs_block(extra)
    assert c["cfg4"] == {"us": 120.0, "frac": 0.83, "GBs": 6666.0, "copy_GBs": 6900.0}
    assert c["cfg3"]["launch_us"] == 7.9 and c["cfg3"]["tick4_us"] == 5.9 and c["cfg3"]["queue_us"] == 4.9 and c["cfg3"]["surfaces"] == 36
    result = {"metric": bench.baseline_metric(), "value": 164000.0, "unit": "Mpix/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.00249, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "w" * 600, "submission": "ticks", "kernel": "k1"},
              "roofline": {"bound": "hbm", "achieved": 3660.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.4576, "traffic": 218972481, "steps_per_launch": 16,
                           "residency": W.residency(96, 7.9e6, 4.9e6), "sweep_us": {"32": 2.43, "48": 2.45, "96": 2.49}, "traffic_src": "x" * 300},
              "cpu_baseline": {"value": 610.0, "unit": "Mpix/s", "cores": 16, "kind": "port", "sample": "s" * 400},
              "ticks_ok": True, "tick64_us_per_step": 2.32, "queue_opt_in": {"us_per_step": 2.6, "frac": 0.44, "sweep_us": {"20": 2.15, "48": 2.4, "96": 2.6}},
              "configs": c}
    text = bench.compact_line(result)
    assert len(text) < bench.COMPACT_LIMIT
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["roofline"]["residency"]["frames"] == 96 and line["roofline"]["sweep_us"]["96"] == 2.49 and line["roofline"]["steps_per_launch"] == 16
    assert line["queue_opt_in"]["sweep_us"]["20"] == 2.15 and line["configs"]["cfg3"]["tick4_us"] == 5.9 and line["ticks_ok"] is True
    assert "not this run" in line["roofline"]["traffic_src"]


def test_every_rank_rotates_over_the_same_number_of_frames():
    """bench_dist.py: the rotation's length fixes the layout of the IPC-shared allocation the peers map (tensors, then flag words).  Sized
    from each rank's OWN crop lists it differed between ranks (round 5: a memory fault at 4 ranks) -- the rule takes no rank."""
    import inspect

    import bench_dist
    assert "rank" not in inspect.signature(bench_dist.rotation_length).parameters
    n, rd, wr = bench_dist.rotation_length(0, 64, 4)
    assert n == W.rotation_units(W.k1_touched_per_frame(64, W.FRAME_6K, 0, 4)[0]) and n * rd >= 2 * W.LLC_BYTES and wr == 64 * 3 * 4 * 64 * 128
    per_rank = {W.rotation_units(W.k1_touched_per_frame(64, W.FRAME_6K, r, 4)[0]) for r in range(8)}
    assert len(per_rank) > 1          # ... which is exactly why: the per-rank figures do differ
    assert bench_dist.rotation_length(5, 64, 4)[0] == 5 and bench_dist.rotation_length(0, 64, 4, one_gpu=True)[0] == 8
