#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries
profiles/<label>_*.txt and refresh profiles/pmc_headline.json.   python tools/make_profile_summaries.py r01c r01_c"""
import csv
import io
import json
import os
import statistics
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prof_summary  # noqa: E402

tag, label = sys.argv[1], sys.argv[2]
R = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def cap(fn, *a):
    buf = io.StringIO()
    with redirect_stdout(buf):
        fn(*a)
    return buf.getvalue()


def pmc_mean(path, sub):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"]]
    return statistics.mean(v) if v else float("nan")


def cut(text, n=200):
    return "\n".join(l[:n] for l in text.splitlines()) + "\n"


bench = json.loads(open(os.path.join(R, "bench_trace.json")).read().strip().splitlines()[-1])
unprof = None
if os.path.exists(os.path.join(R, "bench_unprofiled.json")):
    unprof = json.loads(open(os.path.join(R, "bench_unprofiled.json")).read().strip().splitlines()[-1])
with open(os.path.join(P, label + "_k1_bench_kernel_stats.txt"), "w") as f:
    f.write("# %s: rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu --no-extra\n" % label)
    f.write("# bench line of the SAME (profiled) run: value %s Mpix/s, ms_per_step %s, roofline.kernel_us %s, frac %s\n" % (
        bench["value"], bench["ms_per_step"], bench["roofline"]["kernel_us"], bench["roofline"]["frac"]))
    if unprof:
        f.write("# un-profiled `python bench.py` (separate gpurun call, MI355X box of the same pool): value %s Mpix/s, ms_per_step %s, kernel_us %s, frac %s\n" % (
            unprof["value"], unprof["ms_per_step"], unprof["roofline"]["kernel_us"], unprof["roofline"]["frac"]))
    f.write("# (profiled passes run 10-15 % slower: profiler serialisation + lower clocks, MI355X_MICROARCH.md DVFS note)\n")
    f.write("# rocprofv3's own kernel stats:\n")
    f.write(cut("\n".join(open(os.path.join(R, "bench_trace", "t_kernel_stats.csv")).read().splitlines()[:6]), 230))
    f.write("\n" + cut(cap(prof_summary.kernels, os.path.join(R, "bench_trace", "t_kernel_trace.csv")), 190))

with open(os.path.join(P, label + "_cfg3_cfg4_kernel_stats.txt"), "w") as f:
    f.write("# %s: secondary configs, rocprofv3 --kernel-trace --stats -- python tools/bench_more.py --iters 50\n" % label)
    f.write(open(os.path.join(R, "more_trace.json")).read())
    f.write("\n" + cut(cap(prof_summary.kernels, os.path.join(R, "more_trace", "t_kernel_trace.csv")), 190))

k1 = {}
with open(os.path.join(P, label + "_pmc_hbm.txt"), "w") as f:
    f.write("# %s: HBM counters (KB), separate --pmc passes, and their calibration on known byte counts (tools/calibrate_pmc.py)\n" % label)
    for n in (50, 3200):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(R, "pmc_%s_%d" % (c, n), "p_counter_collection.csv")
            k1[(n, c)] = pmc_mean(path, "k1_resize_split")
            f.write("## K1, crops per launch = %d, %s\n" % (n, c))
            f.write(cut(cap(prof_summary.pmc, path, "k1_resize"), 180))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write("## calibration launches (tools/calibrate_pmc.py), %s\n" % c)
        f.write(open(os.path.join(R, "calib_%s.txt" % c)).read())
        f.write(cut(cap(prof_summary.pmc, os.path.join(R, "calib_%s" % c, "p_counter_collection.csv")), 180))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write("## cfg3 / cfg4 kernels (tools/bench_more.py), %s\n" % c)
        f.write(cut(cap(prof_summary.pmc, os.path.join(R, "more_%s" % c, "p_counter_collection.csv")), 180))
    cal_k1 = pmc_mean(os.path.join(R, "calib_FETCH_SIZE", "p_counter_collection.csv"), "k1_resize_split")
    cal_cp = pmc_mean(os.path.join(R, "calib_FETCH_SIZE", "p_counter_collection.csv"), "k_plane_copy")
    alg50 = bench["roofline"]["algorithmic_bytes_per_launch"]
    f.write("""# Reading (units KB = 1024 B):
#  * WRITE_SIZE is exact on every pattern here (K1 identity launch: 398,131,200 B -> 388,800 KB).
#  * FETCH_SIZE halves 16 B/lane streams: the plane copy reads 373,248,000 B and reports %.0f KB (x%.3f) -- the gfx950
#    under-report of MI355X_MICROARCH.md; the 12 B/lane pointwise kernel is halved as well.
#  * K1's 8 B/lane unaligned taps: the identity launch must read >= 99,532,800 B (+ <= 1/16 for rows shared between
#    16-row tiles) and reports %.0f KB = %.0f B: under-report 1.16x .. 1.24x -> correction x1.2.
#  * traffic(50 crops)  = %.1f KB x 1.2 + %.1f KB = %.2f MB per launch vs %.2f MB algorithmic
#    traffic(3200 crops) = %.0f KB x 1.2 + %.0f KB = %.1f MB per launch
""" % (cal_cp, cal_cp * 1024 / 373248000.0, cal_k1, cal_k1 * 1024, k1[(50, "FETCH_SIZE")], k1[(50, "WRITE_SIZE")],
       (k1[(50, "FETCH_SIZE")] * 1.2 + k1[(50, "WRITE_SIZE")]) * 1024 / 1e6, alg50 / 1e6, k1[(3200, "FETCH_SIZE")],
       k1[(3200, "WRITE_SIZE")], (k1[(3200, "FETCH_SIZE")] * 1.2 + k1[(3200, "WRITE_SIZE")]) * 1024 / 1e6))

json.dump({"source": "profiles/%s_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % label,
           "workload": "cfg2b, 50 crops per launch", "kernel": "k1_resize_split<3,64,1,K1Prog<100,2,4,5>,0>",
           "fetch_size_kb": round(k1[(50, "FETCH_SIZE")], 2), "write_size_kb": round(k1[(50, "WRITE_SIZE")], 2),
           "fetch_correction": 1.2,
           "correction_note": "FETCH_SIZE under-reports K1's 8 B/lane taps by 1.16-1.24x (tools/calibrate_pmc.py); WRITE_SIZE is exact"},
          open(os.path.join(P, "pmc_headline.json"), "w"), indent=1)
print("wrote profiles/%s_*" % label)
