#!/usr/bin/env python3
"""Per-kernel means of the two counter passes of tools/pmc_pair.sh (gpurun_out/pmc_a, pmc_b)."""
import collections
import csv
import glob
import os
for d in ("gpurun_out/pmc_a", "gpurun_out/pmc_b"):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        if any(k.startswith(p) for p in os.environ.get("PMC_KERNELS", "pr_cold_tile,pr_cold_accum").split(",")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x[len(x) // 2:]) / len(x[len(x) // 2:]) / 1e6, 1) for c, x in v.items()}, "(millions)")
