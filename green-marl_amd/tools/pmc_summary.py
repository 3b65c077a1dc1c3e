#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (csv output) of bench.py into per-step HBM traffic.

usage: pmc_summary.py <out.json> <steps_per_run> <dir-or-csv> [<dir-or-csv> ...]

Every pass directory holds a *counter_collection.csv.  Per kernel of the PageRank step (names starting with
pr_), counters are averaged per launch; the step total is sum over kernels of (launches per step x mean).
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE count KiB, and on gfx950
FETCH_SIZE under-reports reads by 2x."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STEP_KERNELS = ["pr_cold_tile_kernel", "pr_cold_accum_kernel", "pr_cold_reduce_few_kernel", "pr_cold_reduce_kernel", "pr_diff_reduce2_kernel",
                "pr_wave_sliced_kernel", "pr_wave_kernel", "pr_sliced_fixup_kernel", "pr_fixup_kernel",
                "pr_combine_kernel", "pr_diff_reduce_kernel"]


def short(name):
    for k in STEP_KERNELS:
        if k + "<" in name or name.startswith(k + "("):
            return k
    return None


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for src in sys.argv[3:]:
        files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            per_dispatch = defaultdict(float)     # (dispatch, kernel, counter) -> value summed over instances
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                per_dispatch[(row["Dispatch_Id"], k, row["Counter_Name"])] += float(row["Counter_Value"])
            for (_, k, c), v in per_dispatch.items():
                acc[k][c][0] += v
                acc[k][c][1] += 1
    res, tot = {}, defaultdict(float)
    for k, cs in acc.items():
        res[k] = {}
        for c, (v, n) in cs.items():
            res[k][c] = v / n
            res[k]["launches_seen_" + c] = n
            tot[c] += v / n        # every step kernel runs once per step (chunks = 1)
    fetch, write = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
    res["_step_total"] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                          "hbm_read_bytes_corrected_x2": 2 * fetch * 1024, "hbm_write_bytes": write * 1024,
                          "hbm_bytes_per_step": (2 * fetch + write) * 1024,
                          "TCC_REQ_sum": tot.get("TCC_HIT_sum", 0.0) + tot.get("TCC_MISS_sum", 0.0),
                          "TCC_MISS_sum": tot.get("TCC_MISS_sum", 0.0)}
    res["_note"] = ("per-launch means of the kernels of one gmx_pr_step, summed over the step; separate rocprofv3 --pmc "
                    "passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum) of `python3 bench.py --no-cpu`; "
                    "first-sweep-only kernels (pr_inactive_*) are not part of a steady-state step")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["_step_total"], indent=1))


if __name__ == "__main__":
    main()
