#!/usr/bin/env python3
"""Development tool: BFS / TC timings at large scales."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx
scales = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "22,24,26").split(",")]
do_tc = len(sys.argv) > 2 and sys.argv[2] == "tc"
for scale in scales:
    for perm in (0, 1):
        g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, bool(perm))
        root = 0
        if perm:
            import numpy as np
            begin = g.download(reverse=False)[0]
            root = int(np.argmax(np.diff(begin)))
        for _ in range(2):
            dist, s = g.hop_dist(root)
        print("RMAT-%d perm=%d root=%d hop_dist %.3f ms levels=%d reached=%d examined=%d E_r=%d  %.1f GTEPS (E_r/t)  roofline(8E_r+12V_r)/t = %.0f GB/s" % (
            scale, perm, root, s["kernel_ms"], s["iterations"], s["vertices_reached"], s["edges_examined"], s["edges_reached"],
            s["edges_reached"] / s["kernel_ms"] / 1e6,
            (8 * s["edges_reached"] + 12 * s["vertices_reached"]) / s["kernel_ms"] / 1e6), flush=True)
        if do_tc and perm == 0:
            t0 = time.time()
            T, s = g.triangle_counting()
            print("RMAT-%d perm=%d triangle_counting(directed) T=%d %.1f ms" % (scale, perm, T, s["kernel_ms"]), flush=True)
        g.free()
