#!/usr/bin/env python3
"""Development tool: per-level counts of one hop_dist traversal (GMX_BFS_DEBUG=1) and, from the result, how many vertices
each bottom-up level had looking for a parent and the in-degrees of the ones that cannot have found one."""
import os, sys
os.environ["GMX_BFS_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gmx
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)
dist, s = g.hop_dist(0)
print("hop_dist %.3f ms levels %d reached %d examined %d" % (s["kernel_ms"], s["iterations"], s["vertices_reached"], s["edges_examined"]), flush=True)
rb = g.download(reverse=True)[2]
indeg = np.diff(rb.astype(np.int64))
del rb
has = indeg > 0
INF = 2147483647
for L in range(1, int(dist[dist != INF].max()) + 2):
    looking = has & (dist >= L)
    fail = looking & (dist > L)
    d = indeg[fail]
    n = max(1, len(d))
    print("level %d: looking %d, found %d, not found %d (indeg 1: %.1f%% 2: %.1f%% 3-4: %.1f%% 5-8: %.1f%% 9-32: %.1f%% >32: %.1f%%), entries in their rows %d" % (
        L, int(looking.sum()), int((looking & (dist == L)).sum()), len(d), 100.0 * (d == 1).sum() / n, 100.0 * (d == 2).sum() / n,
        100.0 * ((d >= 3) & (d <= 4)).sum() / n, 100.0 * ((d >= 5) & (d <= 8)).sum() / n, 100.0 * ((d >= 9) & (d <= 32)).sum() / n,
        100.0 * (d > 32).sum() / n, int(d.sum())), flush=True)
