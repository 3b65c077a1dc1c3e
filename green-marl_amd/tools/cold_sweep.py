#!/usr/bin/env python3
"""Development helper: one RMAT graph, PageRank step time for several hot/cold thresholds (GMX_PR_COLD) and
phase-2 chunk sizes.  usage: cold_sweep.py [scale] [elem] [hot,hot,...]   (hot = ids per rank range; 0 = every edge binned, -1 = no binned part, -2 = library default)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gmx  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
elem = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hots = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20]
steps = int(os.environ.get("SWEEP_STEPS", "10"))
ranks = int(os.environ.get("SWEEP_RANKS", "1"))       # time rank 0 of an N-rank partition (no exchange)
chunks = int(os.environ.get("SWEEP_CHUNKS", "1"))
gmx.require_device()
gmx.set_device(0)
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
base = int(os.environ.get("SWEEP_OPTIONS", gmx.default_pr_options(1 << scale, ranks)))
for hot in hots:
    opts = base
    if hot == -2:   # the library's size rule
        os.environ.pop("GMX_PR_COLD", None)
    elif hot < 0:
        opts &= ~gmx.GMX_PR_COLD_PB
        os.environ.pop("GMX_PR_COLD", None)
    else:
        os.environ["GMX_PR_COLD"] = str(hot)
    t0 = time.perf_counter()
    st = gmx.PageRankState(g, elem, 0, ranks, opts)
    if chunks > 1:
        st.set_chunks(chunks)
    st.reset(0.85)
    plan_s = time.perf_counter() - t0
    for _ in range(3):
        st.step()
    st.timing(True)
    for _ in range(steps):
        st.step()
    n, ms = st.kernel_time()
    info = st.cold_info()
    print("scale %d elem %d ranks %d chunks %d hot %9d: %.3f ms/step  (%d steps; cold edges %d = %.1f %%, padded %d; plan %.1f s; diff %.6e)"
          % (scale, elem, ranks, chunks, hot, ms, n, info["cold_edges"], 100.0 * info["cold_edges"] / g.E, info["padded_items"], plan_s, st.diff()),
          flush=True)
    st.free()
g.free()
