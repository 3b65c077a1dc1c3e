#!/usr/bin/env python3
"""Development tool: what the dist[] download of a hop_dist call costs into a fresh and into a pre-touched host array."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gmx

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)
g.hop_dist(0)
for label in ("fresh", "touched", "fresh", "touched"):
    dist = np.empty(g.V, np.int32)
    if label == "touched":
        dist[::1024] = 0          # one write per 4 KiB page
    st = gmx.Stats()
    t0 = time.perf_counter()
    gmx._ck(gmx.lib().gmx_hop_dist(g._h, 0, dist.ctypes.data, C.byref(st)))
    wall = (time.perf_counter() - t0) * 1e3
    print("%-8s wall %.2f ms, traversal %.3f ms, download %.2f ms (%.1f GB/s)" % (label, wall, st.kernel_ms, st.d2h_ms, 4e-6 * g.V / st.d2h_ms), flush=True)
