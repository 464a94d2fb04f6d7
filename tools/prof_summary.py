#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small text files committed under profiles/.

  prof_summary.py kernels <dir-or-file>     per-kernel count / avg / min / max / total duration
                                            (rocpd .db or *_kernel_trace.csv from --kernel-trace)
  prof_summary.py pmc <counter_collection.csv> [kernel-substring]
                                            per-kernel mean of every collected counter
  prof_summary.py calls <dir-or-file> <kernel-substring>
                                            every call of the matching kernels with its duration -- for the descriptor queue's
                                            server grid, where ONE call serves a whole timed region (bench.py prints how many
                                            batches it served: duration / batches = the per-batch kernel time)
  prof_summary.py pmccalls <counter_collection.csv> <kernel-substring>
                                            every call's counter values (same reason)
"""
import csv
import glob
import os
import sqlite3
import statistics
import sys


def kernel_rows(path):
    if os.path.isdir(path):
        cands = glob.glob(os.path.join(path, "**", "*.db"), recursive=True) + \
            glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        path = cands[0]
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        q = "select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (disp, sym)
        return [(n, int(s), int(e)) for n, s, e in c.execute(q)], path
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return rows, path


def short(name, n=100):
    return name if len(name) <= n else name[:n - 3] + "..."


def kernels(path):
    rows, src = kernel_rows(path)
    by = {}
    for n, s, e in rows:
        by.setdefault(n, []).append(e - s)
    total = sum(sum(v) for v in by.values())
    print("# source: %s" % os.path.basename(src))
    print("%-100s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("%-100s %8d %12d %10.0f %10d %10d %6.2f%%" % (short(n), len(v), sum(v), statistics.mean(v), min(v), max(v),
                                                             100.0 * sum(v) / total))


def calls(path, sub):
    rows, src = kernel_rows(path)
    print("# source: %s" % os.path.basename(src))
    print("%-60s %6s %14s" % ("kernel", "call", "duration_us"))
    i = 0
    for n, s0, e0 in sorted(rows, key=lambda r: r[1]):
        if sub in n:
            print("%-60s %6d %14.1f" % (short(n, 60), i, (e0 - s0) / 1e3))
            i += 1


def pmccalls(path, sub):
    print("# source: %s" % os.path.basename(path))
    print("%-60s %-16s %18s" % ("kernel", "counter", "value"))
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            print("%-60s %-16s %18.1f" % (short(r["Kernel_Name"], 60), r["Counter_Name"], float(r["Counter_Value"])))


def pmc(path, sub=""):
    by = {}
    for r in csv.DictReader(open(path)):
        if sub and sub not in r["Kernel_Name"]:
            continue
        by.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    print("# source: %s" % os.path.basename(path))
    print("%-100s %-16s %8s %16s %16s %16s" % ("kernel", "counter", "calls", "mean", "min", "max"))
    for (n, c), v in sorted(by.items()):
        print("%-100s %-16s %8d %16.3f %16.3f %16.3f" % (short(n), c, len(v), statistics.mean(v), min(v), max(v)))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    if sys.argv[1] == "calls":
        calls(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmccalls":
        pmccalls(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "kernels":
        kernels(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
