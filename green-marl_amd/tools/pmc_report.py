#!/usr/bin/env python3
"""pmc_report.py <tag> -- per-kernel means of the PMC passes of tools/prof_cmd.sh (gpurun_out/<tag>_{fetch,write,sq}) next to
the kernel-trace durations.  HBM bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE count
KiB; on gfx950 FETCH_SIZE under-reports reads by 2x (corrected here)."""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write", "sq"):
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (tag, kind), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
dur = {}
for f in glob.glob("gpurun_out/%s_kernel_stats.csv" % tag):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
rows = []
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rd, wr = 2 * m.get("FETCH_SIZE", 0.0) * 1024, m.get("WRITE_SIZE", 0.0) * 1024
    calls, us = dur.get(k, (0, 0.0))
    rows.append((us * calls, k, calls, us, rd, wr, m))
rows.sort(reverse=True)
print("%-64s %6s %10s %10s %10s %9s %7s %7s %7s" % ("kernel", "calls", "avg us", "read MB", "write MB", "GB/s", "VALU%", "LDS%", "wait%"))
for tot, k, calls, us, rd, wr, m in rows[:24]:
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    pct = lambda c: 100.0 * m.get(c, 0.0) / wc if wc else 0.0   # noqa: E731
    print("%-64s %6d %10.1f %10.1f %10.1f %9.0f %7.1f %7.1f %7.1f" % (k[:64], calls, us, rd / 1e6, wr / 1e6, (rd + wr) / us / 1e3 if us else 0,
                                                                   pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"), pct("SQ_WAIT_INST_ANY")))
