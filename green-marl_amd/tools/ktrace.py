#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV, split into segments at each occurrence of a marker kernel
(default pr_reset_kernel: one segment per plan in tools/cold_sweep.py).  usage: ktrace.py <kernel_trace.csv> [prefix] [marker]"""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
prefix = sys.argv[2] if len(sys.argv) > 2 else "pr_"
marker = sys.argv[3] if len(sys.argv) > 3 else "pr_reset_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
segs, cur = [], None
for r in rows:
    name = r["Kernel_Name"].replace("void ", "")
    if marker in name:
        cur = OrderedDict()
        segs.append(cur)
    if cur is None or not name.startswith(prefix):
        continue
    short = name.split("(")[0].split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cur.setdefault(short, []).append(d)
for i, s in enumerate(segs):
    print("segment %d" % i)
    for k, v in s.items():
        v2 = v[len(v) // 3:] if len(v) > 6 else v     # drop warm-up calls
        print("  %-28s calls %3d  avg %9.1f us  min %9.1f  max %9.1f" % (k, len(v), sum(v2) / len(v2), min(v2), max(v2)))
