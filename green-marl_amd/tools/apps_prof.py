#!/usr/bin/env python3
"""Development tool: device times of the §8f apps (sssp, avg_teen_cnt, conduct, bc) on RMAT-<scale> (default 24)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gmx

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
rng = np.random.default_rng(1)
length = rng.integers(1, 101, g.E).astype(np.int32)
begin = g.download(reverse=False)[0]
root = int(np.argmax(np.diff(begin)))
for _ in range(2):
    dist, st = g.sssp(length, root)
print("RMAT-%d sssp from the top hub: %.2f ms, iterations %d, reached %d" % (scale, st["kernel_ms"], st["iterations"], int((dist != 2147483647).sum())), flush=True)
age = rng.integers(0, 100, g.V).astype(np.int32)
for _ in range(2):
    avg, cnt, st = g.avg_teen_cnt(age, 5)
print("RMAT-%d avg_teen_cnt: %.2f ms (%.1f GTEPS)" % (scale, st["kernel_ms"], g.E / st["kernel_ms"] / 1e6), flush=True)
member = rng.integers(0, 4, g.V).astype(np.int32)
for _ in range(2):
    c, st = g.conduct(member, 1)
print("RMAT-%d conduct: %.2f ms (%.1f GTEPS)" % (scale, st["kernel_ms"], g.E / st["kernel_ms"] / 1e6), flush=True)
seeds = np.array([root, 1, 2, 3, 4], np.int32)
for _ in range(2):
    bc, st = g.bc(seeds)
print("RMAT-%d bc, 5 seeds: %.2f ms" % (scale, st["kernel_ms"]), flush=True)
