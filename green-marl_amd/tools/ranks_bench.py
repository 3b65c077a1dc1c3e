"""ranks_bench.py [scale] -- the compute side of the N-rank PageRank step on ONE GPU: rank 0 of an N-rank partition of
RMAT-<scale> (N = 1, 2, 4, 8), ms per step without any exchange, and the bytes the exchange would move per rank and step
(packed lists against the full prefixes)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gmx  # noqa: E402

gmx.require_device()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
elem = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
ranks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
for n in ranks:
    t0 = time.perf_counter()
    st = gmx.PageRankState(g, elem, 0, n, gmx.default_pr_options(g.V, n))
    build = time.perf_counter() - t0
    st.reset(0.85)
    for _ in range(3):
        st.step()
    st.timing(True)
    for _ in range(10):
        st.step()
    _, ms = st.kernel_time()
    line = "N = %d  rank 0: %.3f ms per step (plan %.2f s), %d edges" % (n, ms, build, st.work()["edges"])
    if n > 1:
        st.set_chunks(2)
        st.timing(True)
        for _ in range(10):
            st.step()
        _, ms2 = st.kernel_time()
        info = st.packed_info()
        need = st.exchange_count()
        line += "; two row chunks %.3f ms; exchange per rank and step: packed %.1f MB out / %.1f MB in, full prefixes %.1f MB; per peer packed %s MB" % (
            ms2, sum(info["send"]) * elem / 1e6, sum(info["recv"]) * elem / 1e6, need * (n - 1) * elem / 1e6,
            "/".join("%.1f" % (c * elem / 1e6) for r, c in enumerate(info["send"]) if r != 0))
    print(line, flush=True)
    st.free()
