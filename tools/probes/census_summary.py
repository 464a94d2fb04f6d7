#!/usr/bin/env python3
"""Folds the per-pass prof_summary tables of tools/profile_r06_a.sh (pmc_requests_<pass>_<workload>.txt) into one census table and turns the
request counts into bytes: read = 128 B x RDREQ_128B + 64 B x RDREQ_64B + 32 B x RDREQ_32B, write = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B).
usage: census_summary.py gpurun_out/r06_a > profiles/r06_a_request_census.txt"""
import glob
import os
import re
import sys

d = sys.argv[1]
rows = {}
for path in sorted(glob.glob(os.path.join(d, "pmc_requests_*_*.txt"))):
    wl = re.sub(r"^pmc_requests_\d+_", "", os.path.basename(path))[:-4]
    for line in open(path):
        m = re.search(r"\s(T[A-Z0-9_]+_sum)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            kern = line.split("(")[0].replace("void cvgs::", "")[:70]
            rows.setdefault(wl, {})[m.group(1)] = (float(m.group(3)), int(m.group(2)), kern)
print("# L2 -> fabric request census (raw TCC / TCP counters, separate rocprofv3 --pmc passes with --kernel-trace only; tools/profile_r06_a.sh).")
print("# ticks16 = the headline launch (bench.py --eager --steps 128 ..., 16 frames x 50 crops per cvgs_execute_many launch, 96-frame rotation);")
print("# calibrate_A / _C / _B = tools/calibrate_pmc.py launches of KNOWN byte counts (K1 at identity scale over four 4K frames; K1 4:1 x 1:1 over the same")
print("# frames; the 16-byte streaming copy of 373,248,000 B).  Means per launch.")
for wl in sorted(rows, key=lambda k: (not k.startswith("ticks"), k)):
    r = rows[wl]
    kern = next(iter(r.values()))[2]
    print("\n== %s   kernel %s" % (wl, kern))
    for c in sorted(r):
        print("%-28s %16.1f   (%d launches)" % (c, r[c][0], r[c][1]))
    g = lambda c: r.get(c, (0.0,))[0]
    rd = 128 * g("TCC_EA0_RDREQ_128B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 32 * g("TCC_EA0_RDREQ_32B_sum")
    wr = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))
    tot = g("TCC_EA0_RDREQ_sum")
    print("-> read %.0f B (%.1f %% of the requests are 128-byte ones, %.2f %% 64-byte, %.2f %% 32-byte); write %.0f B; FETCH_SIZE's formula (TCC_BUBBLE = 0 here) would say %.0f B"
          % (rd, 100 * g("TCC_EA0_RDREQ_128B_sum") / max(tot, 1), 100 * g("TCC_EA0_RDREQ_64B_sum") / max(tot, 1), 100 * g("TCC_EA0_RDREQ_32B_sum") / max(tot, 1), wr,
             64 * (tot - g("TCC_EA0_RDREQ_32B_sum")) + 32 * g("TCC_EA0_RDREQ_32B_sum")))
