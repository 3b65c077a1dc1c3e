#!/usr/bin/env python3
"""Experiment driver (development tool): time graph build, PageRank variants, BFS, TC on the GPU."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gmx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scales", default="20,22,24")
    ap.add_argument("--permute", default="0,1")
    ap.add_argument("--opts", default="0,1,3,5,7")
    ap.add_argument("--elems", default="4")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bfs", type=int, default=1)
    ap.add_argument("--tc", type=int, default=0)
    ap.add_argument("--nranks", type=int, default=1, help="time rank 0 of N ranks (no exchange: timing only)")
    ap.add_argument("--chunks", type=int, default=1, help="row chunks per step (gmx_pr_set_chunks)")
    args = ap.parse_args()
    print(gmx.device_info(), flush=True)
    for scale in [int(s) for s in args.scales.split(",")]:
        for perm in [int(s) for s in args.permute.split(",")]:
            N, M = 1 << scale, 16 << scale
            t0 = time.time()
            g = gmx.Graph.rmat(N, M, 1997, 0.57, 0.19, 0.19, bool(perm))
            t1 = time.time()
            print("RMAT-%d permute=%d: V=%d E=%d build %.2fs" % (scale, perm, g.V, g.E, t1 - t0), flush=True)
            for elem in [int(s) for s in args.elems.split(",")]:
                for opt in [int(s) for s in args.opts.split(",")]:
                    t0 = time.time()
                    st = gmx.PageRankState(g, elem, 0, args.nranks, opt)
                    if args.chunks > 1:
                        st.set_chunks(args.chunks)
                    st.reset(0.85)
                    t1 = time.time()
                    for _ in range(2):
                        st.step()
                    st.diff()
                    t2 = time.time()
                    for _ in range(args.iters):
                        st.step()
                    d = st.diff()
                    t3 = time.time()
                    w = st.work()
                    ms = (t3 - t2) * 1e3 / args.iters
                    print("  pagerank elem=%d opt=%d: plan %.2fs  %.3f ms/iter  %.1f GTEPS  alg %.2f GB  -> %.1f%% of 8 TB/s  (diff %.3e)"
                          % (elem, opt, t1 - t0, ms, w["edges"] / ms / 1e6, w["algorithmic_bytes"] / 1e9,
                             100 * w["algorithmic_bytes"] / (ms * 1e-3) / 8e12, d), flush=True)
                    st.free()
            if args.bfs:
                begin = g.download(reverse=False)[0] if scale <= 24 else None
                roots = [0]
                if begin is not None:
                    roots.append(int(np.argmax(np.diff(begin))))
                for root in roots:
                    dist, s = g.hop_dist(root)
                    dist, s = g.hop_dist(root)
                    print("  hop_dist root=%d: %.3f ms levels=%d reached=%d examined=%d -> %.1f GTEPS(examined)"
                          % (root, s["kernel_ms"], s["iterations"], s["vertices_reached"], s["edges_examined"],
                             s["edges_examined"] / max(s["kernel_ms"], 1e-9) / 1e6), flush=True)
            if args.tc:
                T, s = g.triangle_counting()
                print("  triangle_counting(directed): T=%d %.3f ms" % (T, s["kernel_ms"]), flush=True)
            g.free()


if __name__ == "__main__":
    main()
