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
        # the stepping object (gmx_bfs_*): whole traversal on one rank, and rank 0's share of an 8-rank partition
        for nranks in (1, 8):
            st = gmx.BfsState(g, 0, nranks)
            for rep in range(2):
                st.start(root)
                t0 = time.perf_counter()
                lv = bu = 0
                while True:
                    bu += 1 if st.step_begin() or nranks == 1 and False else 0
                    if st.step_end() == 0:
                        break
                    lv += 1
                dt = (time.perf_counter() - t0) * 1e3
            print("   stepping object nranks=%d (rank 0, no exchange): %.3f ms wall, %d levels, %d partitioned" % (nranks, dt, lv, bu), flush=True)
            st.free()
        if do_tc and perm == 0:
            t0 = time.time()
            T, s = g.triangle_counting()
            print("RMAT-%d perm=%d triangle_counting(directed) T=%d %.1f ms" % (scale, perm, T, s["kernel_ms"]), flush=True)
        g.free()
