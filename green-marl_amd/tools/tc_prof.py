#!/usr/bin/env python3
"""Development tool: triangle counting on symmetrised RMAT-<scale> (default 24), twice (second call uses the cached copy)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)
gs = g.symmetrize()
g.free()
for _ in range(2):
    T, st = gs.triangle_counting()
    print("RMAT-%d symmetrised: E=%d T=%d %.1f ms" % (scale, gs.E, T, st["kernel_ms"]), flush=True)
