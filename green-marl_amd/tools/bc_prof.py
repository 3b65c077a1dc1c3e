#!/usr/bin/env python3
"""Development tool: comp_BC on RMAT-<scale> (default 24), five seeds (the top hub and vertices 1-4), twice."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gmx

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
begin = g.download(reverse=False)[0]
seeds = np.array([int(np.argmax(np.diff(begin))), 1, 2, 3, 4], np.int32)
for _ in range(2):
    bc, st = g.bc(seeds)
    print("RMAT-%d bc, 5 seeds: %.2f ms" % (scale, st["kernel_ms"]), flush=True)
