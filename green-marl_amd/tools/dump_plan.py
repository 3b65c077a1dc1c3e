"""dump_plan.py [scale] [elems] -- build the PageRank plan of RMAT-<scale> (GMX_PR_DEBUG=1 prints its shape) and time 10 steps."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx  # noqa: E402

gmx.require_device()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
elems = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8]
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
import time
for elem in elems:
    t0 = time.perf_counter()
    st = gmx.PageRankState(g, elem, 0, 1, gmx.default_pr_options(g.V, 1))
    print("elem", elem, "plan build %.3f s" % (time.perf_counter() - t0), flush=True)
    st.reset(0.85)
    for _ in range(3):
        st.step()
    st.timing(True)
    for _ in range(10):
        st.step()
    print("elem", elem, st.kernel_time(), st.cold_info(), flush=True)
    st.free()
