#!/usr/bin/env python3
"""Development tool: hop_dist latency at small scales. usage: bfs_small.py <scale> [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)
for i in range(reps):
    dist, s = g.hop_dist(0)
    print("RMAT-%d hop_dist %.3f ms levels=%d" % (scale, s["kernel_ms"], s["iterations"]), flush=True)
g.free()
