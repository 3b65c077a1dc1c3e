"""GPU parity tests proper: the HIP path, called through the C ABI (libgmx.so via ctypes),
against the CPU oracle on the same inputs and against the committed golden fixtures.

Bars: bit-exact for integer work (CSR construction, BFS levels, triangle counts);
PageRank within 1e-6 relative of the fp64 CPU result with the same iteration count
(BASELINE.json north_star), fp64 mode within 1e-12."""
import os

import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647
PR_RTOL_F32 = 1e-6      # north_star: "PageRank ranks match within 1e-6 relative"
PR_RTOL_F64 = 1e-12


@pytest.fixture(scope="module")
def gmx():
    import gmx as m
    m.require_device()
    return m


def rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.abs(b)))


def host_graphs(golden):
    out = []
    for name, c in golden["cases"].items():
        man = golden["manifest"]["hand"].get(name[5:]) if name.startswith("hand_") else golden["manifest"]["rmat"][name]
        out.append((name, c, man))
    return out


def test_device_is_gfx950(gmx):
    info = gmx.device_info()
    assert info["arch"].startswith("gfx950"), info


def test_rmat_generator_bit_compatible(gmx, golden):
    """gmx_graph_create_rmat == reference create_RMAT_graph + do_semi_sort + make_reverse_edges."""
    for name, m in golden["manifest"]["rmat"].items():
        if m["N"] > (1 << 14):
            continue
        g = gmx.Graph.rmat(m["N"], m["M"], m["seed"], *m["abc"], permute=m["permute"])
        begin, node_idx, rb, rn = g.download()
        if name in golden["cases"]:
            c = golden["cases"][name]
            assert np.array_equal(begin, c["begin"]), name
            assert np.array_equal(node_idx, c["node_idx"]), name
            assert np.array_equal(rb, c["r_begin"]) and np.array_equal(rn, c["r_node_idx"]), name
        else:
            og = po.Graph(m["N"], *po.rmat_raw_csr(m["N"], m["M"], m["seed"], *m["abc"], permute=m["permute"])[:2]).prepare()
            assert np.array_equal(begin, og.begin) and np.array_equal(node_idx, og.node_idx), name
            assert np.array_equal(rb, og.r_begin) and np.array_equal(rn, og.r_node_idx), name
        g.free()


def test_upload_builds_reverse_and_sorts(gmx, golden):
    for name, c, _ in host_graphs(golden):
        # raw (unsorted) rows, no reverse given: device must do_semi_sort + make_reverse_edges
        g = gmx.Graph.upload(c["begin"], c["raw_node_idx"], flags=gmx.GMX_GRAPH_SORT_ROWS)
        begin, node_idx, rb, rn = g.download()
        assert np.array_equal(begin, c["begin"]) and np.array_equal(node_idx, c["node_idx"]), name
        assert np.array_equal(rb, c["r_begin"]) and np.array_equal(rn, c["r_node_idx"]), name
        g.free()


def test_pagerank_golden_f64(gmx, golden):
    for name, c, m in host_graphs(golden):
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        rank, st = g.pagerank(0.001, 0.85, 100, np.float64)
        assert st["iterations"] == m["pr_iters"], (name, st)
        assert rel_err(rank, c["rank"]) < PR_RTOL_F64, (name, rel_err(rank, c["rank"]))
        g.free()


def test_pagerank_golden_f32(gmx, golden):
    for name, c, m in host_graphs(golden):
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        rank, st = g.pagerank(0.001, 0.85, 100, np.float32)
        assert st["iterations"] == m["pr_iters"], (name, st)
        assert rel_err(rank, c["rank"]) < PR_RTOL_F32, (name, rel_err(rank, c["rank"]))
        g.free()


@pytest.mark.parametrize("options", [0, 1, 3, 5, 7])
@pytest.mark.parametrize("elem", [4, 8])
def test_pagerank_stepping_variants(gmx, golden, options, elem):
    """Every kernel variant (identity numbering, degree-sorted, LDS hot tile, XCD-sliced, sliced + tile) for 20 fixed iterations."""
    for name in ("rmat10_noperm", "rmat10_perm", "hand_star64", "hand_multi_edge", "hand_empty1"):
        c = golden["cases"][name]
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        og = po.Graph(len(c["begin"]) - 1, c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        want, _, _ = po.pagerank(og, 1e-300, 0.85, 20)
        st = gmx.PageRankState(g, elem, 0, 1, options)
        st.reset(0.85)
        for _ in range(20):
            st.step()
        got = st.download()
        tol = PR_RTOL_F32 if elem == 4 else PR_RTOL_F64
        assert rel_err(got, want) < tol, (name, options, elem, rel_err(got, want))
        st.free()
        g.free()


@pytest.mark.parametrize("scale,permute", [(16, False), (16, True), (18, True)])
def test_pagerank_vs_oracle_larger(gmx, scale, permute):
    og = po.rmat_graph(scale, permute=permute)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    want, it, _ = po.pagerank(og, 0.001, 0.85, 100)
    r64, st64 = g.pagerank(0.001, 0.85, 100, np.float64)
    r32, st32 = g.pagerank(0.001, 0.85, 100, np.float32)
    assert st64["iterations"] == it and st32["iterations"] == it
    assert rel_err(r64, want) < PR_RTOL_F64
    assert rel_err(r32, want) < PR_RTOL_F32
    # run-to-run determinism (no float atomics on the path)
    r32b, _ = g.pagerank(0.001, 0.85, 100, np.float32)
    assert np.array_equal(r32, r32b)
    g.free()


def test_hop_dist_golden(gmx, golden):
    for name, c, m in host_graphs(golden):
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        dist, st = g.hop_dist(m["root"])
        assert np.array_equal(dist, c["dist"]), name
        # top-down only (no reverse CSR on the device)
        g2 = gmx.Graph.upload(c["begin"], c["node_idx"], flags=gmx.GMX_GRAPH_NO_REVERSE)
        dist2, _ = g2.hop_dist(m["root"])
        assert np.array_equal(dist2, c["dist"]), name
        g.free()
        g2.free()


def same_f32(a, b):
    """float32 arrays equal bit for bit, any NaN counting as NaN (its sign / payload is the platform's)."""
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))


def test_common_neighbour_iterator_and_tc_cn(gmx, golden):
    """gmx_common_nbrs / gmx_common_nbr_counts = gm_common_neighbor_iter (items in order, multiplicity of the first
    row; fixtures pinned against the compiled reference class) and triangle counting written with it."""
    for name, c, m in host_graphs(golden):
        if "cn_src" not in c:
            continue
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        og = po.Graph(m["N"], c["begin"].copy(), c["node_idx"].copy(), c["r_begin"].copy(), c["r_node_idx"].copy())
        assert np.array_equal(g.common_nbr_counts(c["cn_src"], c["cn_dst"]), c["cn_counts"]), name
        for s, d in list(zip(c["cn_src"], c["cn_dst"]))[:60]:
            assert np.array_equal(g.common_nbrs(s, d), po.common_nbrs(og, s, d)), (name, int(s), int(d))
        if m.get("tc_cn") is not None:
            assert g.triangle_counting_cn()[0] == m["tc_cn"], name
        g.free()
    for scale, permute in ((14, True), (16, False)):
        og = po.rmat_graph(scale, permute=permute)
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
        assert g.triangle_counting_cn()[0] == po.triangle_counting_cn(og)
        gs = g.symmetrize()
        T, _ = gs.triangle_counting()
        assert gs.triangle_counting_cn()[0] == T                      # the LDS-staged oriented kernel serves both forms
        gs.free()
        g.free()


def test_bfs_object_levels_and_bc_golden(gmx, golden):
    """The device BFS object (gm_bfs_template's role for InBFS / InReverse): levels as the template keeps them
    (short, unvisited -2; the fixtures' dist[] equals gm_bfs_template<short,...> levels, oracle/make_golden.py),
    and comp_BC of bc.gm -- this fork's form (root visited: NaN pattern) and upstream's -- bit for bit against
    the fixtures pinned on the reference's template."""
    for name, c, m in host_graphs(golden):
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        lv, n = g.bfs_levels(m["root"])
        want = np.where(c["dist"] == INT_MAX, -2, c["dist"]).astype(np.int16)
        assert np.array_equal(lv, want) and n == int(want.max()) + 1, name
        if "bc" in c:
            assert same_f32(g.bc(c["bc_seeds"], False)[0], c["bc"]), name
            got, st = g.bc(c["bc_seeds"], True)
            assert same_f32(got, c["bc_skip_root"]), name
            assert st["iterations"] == len(c["bc_seeds"])
        g.free()


@pytest.mark.parametrize("scale,permute", [(12, False), (14, True), (16, False), (18, True)])
def test_bc_vs_oracle_larger(gmx, scale, permute):
    """comp_BC on graphs where the traversal goes bottom-up (the compiled reference drops down edges there,
    oracle/make_golden.py): bit-identical with the oracle restatement, which is pinned on the reference's template
    wherever that records every down edge and cross-checked against a float64 Brandes up to scale 12."""
    og = po.rmat_graph(scale, permute=permute)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    rng = np.random.default_rng(scale)
    seeds = np.concatenate([[int(np.argmax(np.diff(og.begin)))], rng.integers(0, og.N, 4)]).astype(np.int32)
    for skip in (False, True):
        got, st = g.bc(seeds, skip)
        assert same_f32(got, po.bc(og, seeds, skip)), (scale, permute, skip)
    assert not np.isnan(got).any() and float(got.max()) > 0
    g.free()


@pytest.mark.parametrize("scale,permute", [(16, False), (18, False), (18, True), (20, False)])
def test_hop_dist_vs_oracle_larger(gmx, scale, permute):
    og = po.rmat_graph(scale, permute=permute)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    for root in (0, int(np.argmax(np.diff(og.begin))), og.N - 1):
        want = po.bfs_queue(og, root)
        dist, st = g.hop_dist(root)
        assert np.array_equal(dist, want), (scale, permute, root)
        assert st["vertices_reached"] == int((want != INT_MAX).sum())
    g.free()


def test_hop_dist_direction_switches_twice(gmx):
    """Two dense clusters joined by a long path: from a vertex of the first the traversal goes top-down, bottom-up
    (the cluster floods), top-down again along the path, and bottom-up a second time in the far cluster -- the
    candidate bitmap the bottom-up levels hand each other has to be rebuilt after the top-down levels in between.
    Also as N rank states side by side (bitmap slices exchanged by device copies)."""
    rng = np.random.default_rng(7)
    nc, npath = 6000, 60
    V = 2 * nc + npath
    src, dst = [], []
    for base in (0, nc + npath):
        a = rng.integers(0, nc, 24 * nc) + base
        b = rng.integers(0, nc, 24 * nc) + base
        src += [a, b]
        dst += [b, a]
    path = np.arange(nc - 1, nc + npath + 1)          # last vertex of cluster one ... first vertex of cluster two
    src += [path[:-1], path[1:]]
    dst += [path[1:], path[:-1]]
    og = po.graph_from_edges(V, np.concatenate(src).astype(np.int32), np.concatenate(dst).astype(np.int32))
    want = po.bfs_queue(og, 0)
    assert (want != INT_MAX).sum() > 0.95 * V and want.max() > npath       # both clusters reached, through the path
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    dist, st = g.hop_dist(0)
    assert np.array_equal(dist, want)
    g.free()
    for nranks in (2, 3):
        got, levels, exchanges = _bfs_ranks_in_one_process(gmx, og, 0, nranks)
        assert all(np.array_equal(d, want) for d in got)
        assert exchanges >= 3                           # bottom-up levels (one exchange each) in BOTH clusters


@pytest.mark.parametrize("mode", ["plain", "hubs_lds", "hubs_memory"])
def test_hop_dist_hint_encodings(gmx, mode, monkeypatch):
    """The bottom-up levels' hint encodings on graphs small enough for the CPU: plain vertex ids (what a graph of more than
    2^30 vertices gets), hub slots probed in the workgroups' LDS copy (graphs of 2^25 vertices and more: first bottom-up
    level), hub slots probed in memory (their later levels; here: ranges shorter than the threshold, i.e. rank states).
    The options are read when a graph's hints are built and when a level is launched."""
    if mode == "plain":
        monkeypatch.setenv("GMX_BFS_PLAIN_HINTS", "1")
    else:
        monkeypatch.setenv("GMX_BFS_HUB_MIN_V", str(1 << 17) if mode == "hubs_lds" else str(1 << 18))
    for scale, permute in [(18, False), (18, True)]:
        og = po.rmat_graph(scale, permute=permute)
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
        hub = int(np.argmax(np.diff(og.begin)))
        for root in (0, hub, og.N - 1):
            want = po.bfs_queue(og, root)
            dist, st = g.hop_dist(root)
            assert np.array_equal(dist, want), (mode, scale, permute, root)
        g.free()
        # rank states: ranges of V / 2 and V / 3 vertices (below the LDS threshold of "hubs_memory", above it for "hubs_lds" / 2)
        want = po.bfs_queue(og, hub)
        for nranks in (2, 3):
            outs, _, exchanges = _bfs_ranks_in_one_process(gmx, og, hub, nranks)
            assert all(np.array_equal(o, want) for o in outs), (mode, nranks)
            assert exchanges >= 1


def test_hop_dist_bad_root(gmx, golden):
    c = golden["cases"]["rmat6_noperm"]
    g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
    dist, _ = g.hop_dist(-1)
    assert (dist == INT_MAX).all()
    g.free()


def test_triangle_counting_golden(gmx, golden):
    for name, c, m in host_graphs(golden):
        want = m["tc"] if "tc" in m else m["tc_directed"]
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        T, _ = g.triangle_counting()
        assert T == want, (name, T, want)
        g2 = gmx.Graph.upload(c["begin"], c["node_idx"], flags=gmx.GMX_GRAPH_NO_REVERSE)
        T2, _ = g2.triangle_counting()
        assert T2 == want, (name, "forward-only", T2, want)
        g.free()
        g2.free()


@pytest.mark.parametrize("name", ["rmat12_noperm", "rmat14_noperm", "rmat14_perm", "rmat16_noperm"])
def test_triangle_counting_manifest(gmx, golden, name):
    m = golden["manifest"]["rmat"][name]
    og = po.Graph(m["N"], *po.rmat_raw_csr(m["N"], m["M"], m["seed"], *m["abc"], permute=m["permute"])[:2]).prepare()
    if m["tc_directed"] is not None:
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
        assert g.triangle_counting()[0] == m["tc_directed"]
        g.free()
    gs = po.symmetrize(og)
    g = gmx.Graph.upload(gs.begin, gs.node_idx, gs.r_begin, gs.r_node_idx)
    assert g.triangle_counting()[0] == m["tc_symmetrized"]
    g.free()


def test_from_edges_matches_oracle(gmx):
    rng = np.random.default_rng(5)
    V, E = 1000, 20000
    src = rng.integers(0, V, E).astype(np.int32)
    dst = rng.integers(0, V, E).astype(np.int32)
    og = po.graph_from_edges(V, src, dst)
    g = gmx.Graph.from_edges(V, src, dst)
    begin, node_idx, rb, rn = g.download()
    assert np.array_equal(begin, og.begin) and np.array_equal(node_idx, og.node_idx)
    assert np.array_equal(rb, og.r_begin) and np.array_equal(rn, og.r_node_idx)
    g.free()


@pytest.mark.parametrize("nranks,options", [(2, 1), (4, 1), (3, 0), (2, 5), (3, 7), (2, 7), (4, 7), (8, 7)])
def test_pagerank_partitioned_ranks_in_one_process(gmx, nranks, options):
    """The C library's 1-D partition for N ranks, exercised on one GPU: every rank's state lives in
    this process and the all-gather is done with plain device copies (torch), then compared with
    the oracle.  (The collective itself is covered by tests/test_dist_gloo.py on CPU.)"""
    import torch
    og = po.rmat_graph(15, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    want, it, _ = po.pagerank(og, 1e-300, 0.85, 12)
    states = [gmx.PageRankState(g, 8, r, nranks, options) for r in range(nranks)]

    def exchange():
        fulls = [torch.as_tensor(s.contrib_full(), device="cuda") for s in states]
        slices = [torch.as_tensor(s.contrib_slice(), device="cuda") for s in states]
        n = slices[0].numel()
        need = states[0].exchange_count()        # prefix of every range that other ranks can read
        assert all(s.exchange_count() == need for s in states) and 0 <= need <= n
        for dst in fulls:
            for r, src in enumerate(slices):
                dst[r * n:r * n + need].copy_(src[:need])
        torch.cuda.synchronize()

    for s in states:
        s.reset(0.85)
    exchange()
    for _ in range(12):
        for s in states:
            s.step()
        exchange()
    out = np.zeros(og.N)
    for s in states:
        s.download(out)
    assert rel_err(out, want) < PR_RTOL_F64
    assert sum(s.work()["edges"] for s in states) == og.M
    for s in states:
        s.free()
    g.free()


@pytest.mark.parametrize("scale,nranks,chunks,elem", [(15, 2, 4, 8), (15, 4, 3, 8), (15, 1, 8, 8), (18, 2, 8, 8), (18, 8, 4, 4),
                                                       (12, 2, 4, 8), (15, 3, 2, 8), (16, 5, 3, 4)])
def test_pagerank_row_chunked_steps(gmx, scale, nranks, chunks, elem):
    """gmx_pr_step_chunk: a sweep enqueued as C row chunks, each chunk's piece exchanged (device copies
    standing in for the all-gather) before the next chunk is computed -- the order of events of the
    overlapped N > 1 step.  Ranks, diff and iteration behaviour must equal the unchunked run / the oracle."""
    import torch
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    iters = 8
    want, it, want_diff = po.pagerank(og, 1e-300, 0.85, iters)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED
    states = [gmx.PageRankState(g, elem, r, nranks, options) for r in range(nranks)]
    for s in states:
        assert s.set_chunks(chunks) == chunks
    ranges = [states[0].chunk_range(c) for c in range(chunks)]
    need = states[0].exchange_count()
    assert all([s.chunk_range(c) for c in range(chunks)] == ranges for s in states)
    # processing order walks the exchanged prefix from the back; together the pieces tile [0, need)
    assert ranges[-1][0] == 0 and sum(cnt for _, cnt in ranges) == need
    assert all(ranges[c + 1][0] + ranges[c + 1][1] == ranges[c][0] for c in range(chunks - 1))
    assert ranges[0][0] + ranges[0][1] == need
    for s in states:
        s.reset(0.85)
    n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()
    fulls = [torch.as_tensor(s.contrib_full(), device="cuda") for s in states]
    for dst in fulls:
        for r, src in enumerate(fulls):
            dst[r * n:r * n + need].copy_(src[r * n:r * n + need])
    torch.cuda.synchronize()
    for _ in range(iters):
        nxt = [torch.as_tensor(s.contrib_next_full(), device="cuda") for s in states]
        for c, (off, cnt) in enumerate(ranges):
            for s in states:
                s.step_chunk(c)
            for dst in nxt:
                for r, src in enumerate(nxt):
                    if dst is not src:
                        dst[r * n + off:r * n + off + cnt].copy_(src[r * n + off:r * n + off + cnt])
        torch.cuda.synchronize()
    out = np.zeros(og.N, dtype=np.float64 if elem == 8 else np.float32)
    for s in states:
        s.download(out)
    assert rel_err(out, want) < (PR_RTOL_F64 if elem == 8 else PR_RTOL_F32)
    diff = sum(s.diff() for s in states)
    assert abs(diff - want_diff) <= (1e-9 if elem == 8 else 1e-3) * max(want_diff, 1e-30) + 1e-15
    for s in states:
        s.free()
    g.free()


@pytest.mark.parametrize("scale,ranks,exchange", [(15, 3, "peer"), (18, 2, "peer"), (21, 4, "peer"), (21, 2, "peer"), (22, 8, "peer"),
                                                  (16, 1, "allgather"), (16, 1, "allreduce")])
def test_pagerank_entry_drives_several_ranks_from_one_thread(gmx, scale, ranks, exchange, monkeypatch):
    """gmx_pagerank_f64 / _f32 over N rank states from one host thread (gmx_pr_multi.hip: per-rank streams, peer
    copies behind every sweep, events as the barrier, diff summed in rank order).  The box has one GPU, so the ranks
    share it (GMX_PR_RANKS); the RCCL forms of the exchange need one rank per device and are run with a single rank
    (library loading, communicator, group calls).  Results: the oracle's, iteration count included.  From 2^21
    vertices on every in-edge is binned and the peer form can run pipelined (GMX_PR_MULTI_PIPELINE=1: two row chunks,
    the tail chunk's copies under the next iteration's hub tiles, events as barriers): the same bits."""
    og = po.rmat_graph(scale, permute=True) if scale <= 18 else None
    if og is None:
        g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, True)
        begin, node_idx, rb, rn = g.download()
        og = po.Graph(1 << scale, begin, node_idx, rb, rn)
    else:
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    want, it, want_diff = po.pagerank(og, 0.001, 0.85, 100)
    monkeypatch.setenv("GMX_PR_RANKS", str(ranks))
    monkeypatch.setenv("GMX_EXCHANGE", exchange)
    for dt, tol in ((np.float64, PR_RTOL_F64), (np.float32, PR_RTOL_F32)):
        rank, st = g.pagerank(0.001, 0.85, 100, dt)
        assert st["iterations"] == it, (dt, st, it)
        assert rel_err(rank, want) < tol
        rank2, st2 = g.pagerank(0.001, 0.85, 100, dt)          # the cached plan, and run-to-run identical
        assert np.array_equal(rank, rank2) and st2["last_diff"] == st["last_diff"]
    if scale >= 21 and exchange == "peer":
        piped = gmx.Graph.upload(*g.download())                # (a fresh device graph: no cached multi-rank plan)
        monkeypatch.setenv("GMX_PR_MULTI_PIPELINE", "1")
        rank3, st3 = piped.pagerank(0.001, 0.85, 100, np.float32)
        assert np.array_equal(rank3, rank2) and st3["iterations"] == st2["iterations"]
        piped.free()
    g.free()


SMALL_SHAPES_SCRIPT = r"""
import os, sys
import numpy as np
sys.path[:0] = [os.path.join(ROOT, "green-marl_amd"), os.path.join(ROOT, "oracle")]
import gmx
import pyoracle as po
gmx.require_device()
assert gmx.LIB_PATH.endswith("libgmx_dbg.so")
def check(og, tag):
    want, it, _ = po.pagerank(og, 0.001, 0.85, 100)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-6)):
        rank, st = g.pagerank(0.001, 0.85, 100, dt)                     # whole-kernel entry, default options
        assert st["iterations"] == it and np.max(np.abs(rank - want) / want) < tol, (tag, dt, st)
    want8 = po.pagerank(og, 1e-300, 0.85, 8)[0]
    for opts in (1, 3, 7, 15):                                          # plain, LDS tile, sliced pull, binned
        for elem in (8, 4):
            if opts == 15:
                os.environ["GMX_PR_COLD"] = "0"
            st = gmx.PageRankState(g, elem, 0, 1, opts)
            st.reset(0.85)
            for _ in range(8):
                st.step()
            err = np.max(np.abs(st.download() - want8) / want8)
            assert err < (1e-12 if elem == 8 else 1e-6), (tag, opts, elem, err)
            st.free()
    g.free()
# the shape of the round-1 fault: RMAT-12 without the final permutation, fp64, whole-kernel entry
check(po.rmat_graph(12, permute=False), "rmat12")
# one merge-path block in all (rows + edges <= 512)
src = np.arange(100, dtype=np.int32); dst = (src * 7 + 3) % 100
check(po.graph_from_edges(100, np.concatenate([src, src]), np.concatenate([dst.astype(np.int32), ((dst + 1) % 100).astype(np.int32)])), "one_block")
# the last block ends exactly on the last row end: rows + edges a multiple of 512
n = 128; e = 3 * 512 - n
rng = np.random.default_rng(3)
s2 = rng.integers(0, n, e).astype(np.int32); d2 = np.concatenate([rng.integers(0, n - 1, e - 1), [n - 1]]).astype(np.int32)
check(po.graph_from_edges(n, s2, d2), "block_ends_on_last_row")
check(po.rmat_graph(15, permute=True), "rmat15")
print("BOUNDS OK")
"""


def test_pagerank_small_shapes_with_bounds_checks(gmx, tmp_path):
    """Round 1 had one `Memory access fault by GPU` in `bin/gmx_bench` (RMAT-12, gmx_pagerank_f64, the unsliced
    wave-worker kernel) while those kernels were being written; it did not come back and its cause was not
    recorded.  This runs exactly that shape -- plus a graph that is a single merge-path block, one whose
    last block ends on the last row, and every kernel variant -- on a build whose wave workers check each index
    they derive from the block table (make debug: -DGMX_PR_BOUNDS, a violation traps and kills the process)."""
    import subprocess
    import sys
    from conftest import ROOT
    pkg = os.path.join(ROOT, "green-marl_amd")
    subprocess.check_call(["make", "-C", pkg, "-j8", "debug"], stdout=subprocess.DEVNULL)
    script = tmp_path / "small_shapes.py"
    script.write_text("ROOT = %r\n" % ROOT + SMALL_SHAPES_SCRIPT)
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, GMX_LIB=os.path.join(pkg, "libgmx_dbg.so")),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "BOUNDS OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("scale,nranks,chunks,elem,hot,chunk2", [
    (18, 1, 1, 4, 16384, 0), (18, 1, 1, 8, 16384, 0), (18, 1, 2, 4, 32768, 8), (18, 1, 1, 8, 16384, 16),
    (18, 4, 1, 8, 16384, 0), (19, 2, 3, 4, 49152, 64), (18, 3, 2, 8, 16384, 8), (17, 1, 1, 4, 16384, 8),
    (18, 1, 1, 4, 0, 0), (18, 1, 2, 8, 0, 8), (16, 2, 1, 8, 0, 0), (19, 4, 2, 4, 0, 16), (14, 1, 1, 4, 0, 0)])
def test_pagerank_cold_sources_binned(gmx, scale, nranks, chunks, elem, hot, chunk2, monkeypatch):
    """GMX_PR_COLD_PB: the in-edges whose source lies past the hot prefix of its rank range leave the pull sweep
    and are summed by the two binned phases (tile-major gather from LDS, bin-major fixed-point accumulation in
    LDS, chunk reduction for split bins).  Small graphs take the path through GMX_PR_COLD (hot ids per rank
    range) and GMX_PR_COLD_CHUNK (groups per phase-2 item, to split bins); rank ranges smaller than an LDS tile
    make tiles straddle rank ranges.  Ranks equal the oracle's, and two runs are bit-identical."""
    import torch
    monkeypatch.setenv("GMX_PR_COLD", str(hot))
    if chunk2:
        monkeypatch.setenv("GMX_PR_COLD_CHUNK", str(chunk2))
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    iters = 6
    want, it, want_diff = po.pagerank(og, 1e-300, 0.85, iters)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | gmx.GMX_PR_COLD_PB
    states = [gmx.PageRankState(g, elem, r, nranks, options) for r in range(nranks)]
    info = [s.cold_info() for s in states]
    assert all(i["hot_ids"] == hot and i["cold_edges"] > 0 and i["padded_items"] > 0 for i in info), info
    if hot == 0:
        assert sum(i["cold_edges"] for i in info) == og.M   # no edge is left to the pull sweep
    assert sum(s.work()["edges"] for s in states) == og.M
    for s in states:
        assert s.set_chunks(chunks) == chunks
    ranges = [states[0].chunk_range(c) for c in range(chunks)]
    need = states[0].exchange_count()
    n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()

    def run():
        for s in states:
            s.reset(0.85)
        fulls = [torch.as_tensor(s.contrib_full(), device="cuda") for s in states]
        for dst in fulls:
            for r, src in enumerate(fulls):
                if dst is not src:
                    dst[r * n:r * n + need].copy_(src[r * n:r * n + need])
        torch.cuda.synchronize()
        for _ in range(iters):
            nxt = [torch.as_tensor(s.contrib_next_full(), device="cuda") for s in states]
            for c, (off, cnt) in enumerate(ranges):
                for s in states:
                    s.step_chunk(c)
                for dst in nxt:
                    for r, src in enumerate(nxt):
                        if dst is not src:
                            dst[r * n + off:r * n + off + cnt].copy_(src[r * n + off:r * n + off + cnt])
            torch.cuda.synchronize()
        out = np.zeros(og.N, dtype=np.float64 if elem == 8 else np.float32)
        for s in states:
            s.download(out)
        return out, sum(s.diff() for s in states)

    out, diff = run()
    assert rel_err(out, want) < (PR_RTOL_F64 if elem == 8 else PR_RTOL_F32)
    assert abs(diff - want_diff) <= (1e-9 if elem == 8 else 1e-3) * max(want_diff, 1e-30) + 1e-15
    out2, diff2 = run()
    assert np.array_equal(out, out2) and diff == diff2          # integer accumulation: run-to-run bit-identical
    for s in states:
        s.free()
    g.free()


@pytest.mark.parametrize("density", [0, 64, 4096])
@pytest.mark.parametrize("scale,nranks,elem", [(18, 1, 4), (18, 1, 8), (19, 2, 4), (17, 3, 8)])
def test_pagerank_binned_forms_agree(gmx, scale, nranks, elem, density, monkeypatch):
    """The tile-major stream has three storage forms (gmx_pr_cold.hip): length classes for every tile (the default),
    round 2's generic pair / edge forms (GMX_PR_COLD_CLASS_DENSITY=0), and the mix (classes only where the (tile, bin)
    cells are dense enough).  Each must match the oracle; the plans differ in tile size and item order, so the fp32 ranks
    may differ between them in the last bits, the fp64 ones agree to 1e-12."""
    import torch
    monkeypatch.setenv("GMX_PR_COLD", "0")
    monkeypatch.setenv("GMX_PR_COLD_CLASS_DENSITY", str(density))
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    iters = 5
    want, _, _ = po.pagerank(og, 1e-300, 0.85, iters)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | gmx.GMX_PR_COLD_PB
    states = [gmx.PageRankState(g, elem, r, nranks, options) for r in range(nranks)]
    need = states[0].exchange_count()
    n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()
    for s_ in states:
        s_.reset(0.85)

    def exchange(fulls):
        for dst in fulls:
            for r, src in enumerate(fulls):
                if dst is not src:
                    dst[r * n:r * n + need].copy_(src[r * n:r * n + need])
        torch.cuda.synchronize()
    exchange([torch.as_tensor(s_.contrib_full(), device="cuda") for s_ in states])
    for _ in range(iters):
        for s_ in states:
            s_.step()
        exchange([torch.as_tensor(s_.contrib_full(), device="cuda") for s_ in states])
    out = np.zeros(og.N, dtype=np.float64 if elem == 8 else np.float32)
    for s_ in states:
        s_.download(out)
        s_.free()
    g.free()
    assert rel_err(out, want) < (PR_RTOL_F64 if elem == 8 else PR_RTOL_F32)


@pytest.mark.parametrize("scale,nranks,elem", [(18, 4, 4), (17, 2, 8), (19, 8, 4), (16, 3, 8), (22, 2, 4)])
def test_pagerank_pipelined_gather_order(gmx, scale, nranks, elem, monkeypatch):
    """The pipelined form of the pushed step (gmx_pr_step_gather): phase 1 over the class-0 tiles may run while the
    peers' TAIL chunks of the previous step are still travelling.  N rank states in one process; the tail pieces
    are withheld -- the peers' copies of them hold NaN -- until every rank's gather(0) has run, then delivered
    before gather(1).  A class-0 tile that read a single live tail source would poison the ranks."""
    import torch
    monkeypatch.setenv("GMX_PR_COLD", "0")
    monkeypatch.setenv("GMX_PR_COLD_CHUNK", "16")
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    iters = 6
    want, it, want_diff = po.pagerank(og, 1e-300, 0.85, iters)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | gmx.GMX_PR_COLD_PB
    states = [gmx.PageRankState(g, elem, r, nranks, options) for r in range(nranks)]
    assert all(s.gather_classes() == 2 for s in states)
    for s in states:
        assert s.set_chunks(2) == 2
        s.reset(0.85)
    if scale >= 22:     # hub pieces wider than an LDS tile: there ARE tiles of hub sources only
        assert all(s.gather_items(0) > 0 and s.gather_items(1) > 0 for s in states)
    (t_off, t_cnt), (h_off, h_cnt) = states[0].chunk_range(0), states[0].chunk_range(1)   # tail is processed first
    assert h_off == 0 and t_off == h_cnt and t_cnt > 0 and h_cnt > 0
    n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()

    def deliver(fulls, off, cnt):
        for dst in fulls:
            for r, src in enumerate(fulls):
                if dst is not src:
                    dst[r * n + off:r * n + off + cnt].copy_(src[r * n + off:r * n + off + cnt])

    def withhold(fulls, off, cnt):
        for q, dst in enumerate(fulls):
            for r in range(nranks):
                if r != q:
                    dst[r * n + off:r * n + off + cnt] = float("nan")

    cur = [torch.as_tensor(s.contrib_full(), device="cuda") for s in states]
    deliver(cur, h_off, h_cnt)
    withhold(cur, t_off, t_cnt)
    torch.cuda.synchronize()
    for _ in range(iters):
        nxt = [torch.as_tensor(s.contrib_next_full(), device="cuda") for s in states]
        for s in states:
            s.step_gather(0)
        torch.cuda.synchronize()
        deliver(cur, t_off, t_cnt)            # the tail chunk of the previous step lands only now
        for s in states:
            s.step_gather(1)
        for s in states:
            s.step_chunk(0)
        withhold(nxt, t_off, t_cnt)           # ... and this step's stays away until the next gather(0) has run
        for s in states:
            s.step_chunk(1)
        deliver(nxt, h_off, h_cnt)
        torch.cuda.synchronize()
        cur = nxt
    out = np.zeros(og.N, dtype=np.float64 if elem == 8 else np.float32)
    for s in states:
        s.download(out)
    diff = sum(s.diff() for s in states)
    assert rel_err(out, want) < (PR_RTOL_F64 if elem == 8 else PR_RTOL_F32)
    assert abs(diff - want_diff) <= (1e-9 if elem == 8 else 1e-3) * max(want_diff, 1e-30) + 1e-15
    for s in states:
        s.free()
    g.free()


@pytest.mark.parametrize("scale,nranks,chunks,elem", [(16, 2, 1, 8), (18, 4, 2, 4), (17, 3, 2, 8), (18, 8, 2, 4), (20, 8, 1, 4)])
def test_pagerank_packed_exchange_in_one_process(gmx, scale, nranks, chunks, elem, monkeypatch):
    """"Send only what is read" (gmx_pr_push_packed / gmx_pr_unpack): N rank states of one process wired to each other's
    landing zones, stepped chunk by chunk with the library's own pack -> copy -> unpack, against the same states
    exchanging the full prefixes by hand.  The ranks must be bit-identical (an unread position never enters a sum), the
    two views of every list must agree (what r sends to q is what q expects from r), and on RMAT the packed volume must
    be well below the full one from 4 ranks up."""
    import torch
    monkeypatch.setenv("GMX_PR_COLD", "0")
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | gmx.GMX_PR_COLD_PB
    iters = 5

    def make():
        states = [gmx.PageRankState(g, elem, r, nranks, options) for r in range(nranks)]
        for s_ in states:
            assert s_.set_chunks(chunks) == chunks
        return states

    def finish(states):
        out = np.zeros(og.N, dtype=np.float64 if elem == 8 else np.float32)
        for s_ in states:
            s_.download(out)
        d = sum(s_.diff() for s_ in states)
        for s_ in states:
            s_.free()
        return out, d

    # (1) the reference run: full prefixes copied by hand
    states = make()
    need = states[0].exchange_count()
    n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()
    for s_ in states:
        s_.reset(0.85)

    def full_exchange():
        fulls = [torch.as_tensor(s_.contrib_full(), device="cuda") for s_ in states]
        for dst in fulls:
            for r, src in enumerate(fulls):
                if dst is not src:
                    dst[r * n:r * n + need].copy_(src[r * n:r * n + need])
        torch.cuda.synchronize()
    full_exchange()
    for _ in range(iters):
        for s_ in states:
            s_.step()
        full_exchange()
    want, want_diff = finish(states)
    # (2) the packed run
    states = make()
    infos = [s_.packed_info() for s_ in states]
    assert all(i is not None for i in infos)
    for r in range(nranks):
        for q in range(nranks):
            assert infos[r]["send"][q] == infos[q]["recv"][r], (r, q)      # both ends hold the same list length
            if r != q:                                                      # ... and the same positions
                a = torch.as_tensor(states[q].recv_list(r), device="cuda")
                assert a.numel() == infos[r]["send"][q] and bool((a[1:] > a[:-1]).all()) and (a.numel() == 0 or int(a[-1]) < need)
    for s_ in states:
        s_.set_peers_local(states)
        assert s_.packed_push
        s_.reset(0.85)
    for s_ in states:
        s_.push_current()
        s_.push_join()
    torch.cuda.synchronize()
    for s_ in states:
        s_.unpack(-1)
    torch.cuda.synchronize()
    for _ in range(iters):
        for c in range(chunks):
            for s_ in states:
                s_.step_chunk(c)
                s_.push_chunk(c)
        for s_ in states:
            s_.push_join()
        torch.cuda.synchronize()              # the per-step barrier of this single-process rehearsal
        for c in range(chunks):
            for s_ in states:
                s_.unpack(c)
        torch.cuda.synchronize()
    sent = sum(s_.exchange_bytes() for s_ in states)
    full = nranks * (nranks - 1) * need * elem
    got, got_diff = finish(states)
    g.free()
    assert np.array_equal(got, want) and got_diff == want_diff
    assert sent <= full
    if nranks >= 4 and scale >= 18:
        assert sent < 0.8 * full, (sent, full)


@pytest.mark.parametrize("world,chunks,elem,binned", [(3, 2, 8, ""), (2, 1, 4, ""), (3, 2, 4, "binned"), (2, 2, 8, "binned"), (2, 1, 4, "binned"),
                                                      (3, 2, 4, "binned-plain"), (2, 1, 8, "plain")])
def test_peer_push_exchange_between_processes(gmx, world, chunks, elem, binned):
    """The N > 1 exchange by direct copies into the peers' hipIpc-mapped replicas, with real processes (one
    per rank, sharing this box's single GPU; see tests/mp_push_worker.py).  binned: plans with every in-edge
    binned, which DistPageRank drives in the pipelined order when the step has two chunks."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mp_push_worker.py")
    procs = []
    plain = "plain" in binned            # the full-prefix push; otherwise the packed one (the default when the plans have lists)
    binned = "binned" if "binned" in binned else ""
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GMX_PUSH_PACKED="0" if plain else "1")
        procs.append(subprocess.Popen([sys.executable, worker, "15", str(chunks), str(elem), "8", binned], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "-> OK" in outs[0], outs[0]


def _bfs_ranks_in_one_process(gmx, og, root, nranks):
    """N rank states of the partitioned hop_dist side by side; device copies stand in for the all-gather."""
    import torch
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    states = [gmx.BfsState(g, r, nranks) for r in range(nranks)]
    for s in states:
        s.start(root)
    levels = exchanges = 0
    while True:
        needs = [s.step_begin() for s in states]
        assert len(set(needs)) == 1                       # every rank takes the same direction
        if needs[0]:
            views = []
            for s in states:
                words, off, n = s.found_bitmap()
                views.append((torch.as_tensor(words, device="cuda"), off, n))
            for dst, _, _ in views:
                for src, off, n in views:
                    if dst is not src:
                        dst[off:off + n].copy_(src[off:off + n])
            torch.cuda.synchronize()
            exchanges += 1
        counts = [s.step_end() for s in states]
        assert len(set(counts)) == 1                      # ... and sees the same next frontier
        if counts[0] == 0:
            break
        levels += 1
    outs = [s.download()[0] for s in states]
    for s in states:
        s.free()
    g.free()
    return outs, levels, exchanges


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_hop_dist_partitioned_ranks_in_one_process(gmx, nranks):
    """gmx_bfs_*: replicated top-down levels, bottom-up levels partitioned by destination range with a
    found-bitmap slice per rank.  dist[] must be bit-exact on every rank."""
    for scale, permute, root in [(16, False, 0), (14, True, 5), (10, False, 3)]:
        og = po.rmat_graph(scale, permute=permute)
        want = po.hop_dist(og, root)[0]
        outs, levels, exchanges = _bfs_ranks_in_one_process(gmx, og, root, nranks)
        for o in outs:
            assert np.array_equal(o, want)
        if scale == 16 and nranks > 1:
            assert exchanges >= 1                           # the hub-rooted traversal does go bottom-up
    # shapes the direction rule treats differently: a chain (always top-down), a star (one huge level),
    # an unreachable root and a root outside the graph
    n = 3000
    chain = po.graph_from_edges(n, np.arange(n - 1, dtype=np.int32), np.arange(1, n, dtype=np.int32))
    star = po.graph_from_edges(n, np.zeros(n - 1, np.int32), np.arange(1, n, dtype=np.int32))
    for og, root in [(chain, 0), (chain, n - 1), (star, 0), (star, 7), (star, n + 5)]:
        want = po.hop_dist(og, root)[0] if root < n else np.full(n, INT_MAX, np.int32)
        outs, _, _ = _bfs_ranks_in_one_process(gmx, og, root, nranks)
        for o in outs:
            assert np.array_equal(o, want)


@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_triangle_counting_parts_add_up(gmx, nparts):
    og = po.rmat_graph(14, permute=True)
    for graph in (og, po.symmetrize(og)):
        g = gmx.Graph.upload(graph.begin, graph.node_idx, graph.r_begin, graph.r_node_idx)
        full = g.triangle_counting()[0]
        assert full == po.triangle_counting(graph)
        assert sum(g.triangle_counting(p, nparts)[0] for p in range(nparts)) == full
        g.free()
    # forward-only form (no reverse CSR on the device)
    g = gmx.Graph.upload(og.begin, og.node_idx, None, None, flags=gmx.GMX_GRAPH_NO_REVERSE)
    assert sum(g.triangle_counting(p, nparts)[0] for p in range(nparts)) == po.triangle_counting(og)
    g.free()


@pytest.mark.parametrize("hubs", ["0", "64", "1024", "1000000"])
def test_triangle_counting_hub_matrix_sizes(gmx, hubs, monkeypatch):
    """The degree-ordered copy keeps the adjacency among its H highest vertices as a bit matrix (slots whose neighbour is a
    hub are counted by bit probes from the tail's side): no hubs, a few, many, and every vertex a hub give the oracle's
    count -- whole and dealt to three parts -- on RMAT graphs and on a clique (every list longer than the LDS slice)."""
    monkeypatch.setenv("GMX_TC_HUBS", hubs)     # read when the copy is built
    for scale in (10, 14, 16):
        sym = po.symmetrize(po.rmat_graph(scale, permute=True))
        g = gmx.Graph.upload(sym.begin, sym.node_idx, sym.r_begin, sym.r_node_idx)
        want = po.triangle_counting(sym)
        assert g.triangle_counting()[0] == want, (hubs, scale)
        assert sum(g.triangle_counting(p, 3)[0] for p in range(3)) == want
        g.free()
    n = 1500
    iu, ju = np.triu_indices(n, 1)
    g = gmx.Graph.from_edges(n, np.concatenate([iu, ju]).astype(np.int32), np.concatenate([ju, iu]).astype(np.int32))
    assert g.triangle_counting()[0] == n * (n - 1) * (n - 2) // 6
    g.free()


def test_triangle_counting_degree_oriented_path(gmx, monkeypatch):
    """Symmetric simple graphs are counted on a degree-ordered copy (T is numbering-independent there); the
    count must equal the emitted-order count and the oracle's.  Symmetric graphs WITH duplicate slots or
    self-loops, and directed graphs, must keep the emitted order (multiplicity rule)."""
    for scale in (10, 14, 16):
        sym = po.symmetrize(po.rmat_graph(scale, permute=True))
        g = gmx.Graph.upload(sym.begin, sym.node_idx, sym.r_begin, sym.r_node_idx)
        fast = g.triangle_counting()[0]
        parts = sum(g.triangle_counting(p, 3)[0] for p in range(3))
        g.free()
        monkeypatch.setenv("GMX_TC_NO_ORIENT", "1")
        g = gmx.Graph.upload(sym.begin, sym.node_idx, sym.r_begin, sym.r_node_idx)
        plain = g.triangle_counting()[0]
        g.free()
        monkeypatch.delenv("GMX_TC_NO_ORIENT")
        monkeypatch.setenv("GMX_TC_NO_LDS", "1")
        g = gmx.Graph.upload(sym.begin, sym.node_idx, sym.r_begin, sym.r_node_idx)
        mem = g.triangle_counting()[0]
        g.free()
        monkeypatch.delenv("GMX_TC_NO_LDS")
        assert fast == plain == mem == parts == po.triangle_counting(sym)
    # a clique larger than the staged-list capacity (3072): upper lists of up to 3299 entries
    n = 3300
    iu, ju = np.triu_indices(n, 1)
    src = np.concatenate([iu, ju]).astype(np.int32)
    dst = np.concatenate([ju, iu]).astype(np.int32)
    g = gmx.Graph.from_edges(n, src, dst)
    assert g.triangle_counting()[0] == n * (n - 1) * (n - 2) // 6
    g.free()
    # symmetric but not simple: every edge twice, plus self-loops
    rng = np.random.default_rng(11)
    V = 300
    a = rng.integers(0, V, 4000).astype(np.int32)
    b = rng.integers(0, V, 4000).astype(np.int32)
    src = np.concatenate([a, b, a, b, np.arange(20, dtype=np.int32)])
    dst = np.concatenate([b, a, b, a, np.arange(20, dtype=np.int32)])
    og = po.graph_from_edges(V, src, dst)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    assert g.triangle_counting()[0] == po.triangle_counting(og)
    g.free()


def test_sssp_golden_and_oracle(gmx, golden):
    """gmx_sssp against the reference-pinned fixtures (every committed case) and the oracle on larger graphs
    with adversarial lengths; unreachable = INT_MAX; bit-exact."""
    for name, c in golden["cases"].items():
        man = golden["manifest"]["rmat"].get(name) or golden["manifest"]["hand"].get(name[len("hand_"):])
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        dist, st = g.sssp(c["sssp_len"], man["root"])
        assert np.array_equal(dist, c["sssp_dist"]), name
        g.free()
    rng = np.random.default_rng(99)
    for scale, permute, hi in [(14, False, 101), (16, True, 101), (15, True, 2), (13, False, 100000)]:
        og = po.rmat_graph(scale, permute=permute)
        length = rng.integers(1, hi, og.M).astype(np.int32)
        root = int(np.argmax(np.diff(og.begin)))
        want = po.sssp(og, length, root)[0]
        g = gmx.Graph.upload(og.begin, og.node_idx, None, None, flags=gmx.GMX_GRAPH_NO_REVERSE)
        dist, st = g.sssp(length, root)
        assert np.array_equal(dist, want), (scale, permute, hi)
        assert st["iterations"] >= 1
        # all lengths 1: sssp == hop_dist
        ones = np.ones(og.M, np.int32)
        assert np.array_equal(g.sssp(ones, root)[0], po.hop_dist(og, root)[0])
        # root outside the graph / isolated vertices
        assert np.all(g.sssp(length, og.N + 3)[0] == INT_MAX)
        g.free()
    # a long chain with a shortcut that only pays off late (many rounds, vertices re-enter the queue)
    n = 2000
    src = np.concatenate([np.arange(n - 1), [0]]).astype(np.int32)
    dst = np.concatenate([np.arange(1, n), [n // 2]]).astype(np.int32)
    og = po.graph_from_edges(n, src, dst)
    length = np.ones(og.M, np.int32)
    srcs = np.repeat(np.arange(n), np.diff(og.begin))
    length[(srcs == 0) & (og.node_idx == n // 2)] = 900          # 0 -> n/2 directly costs 900, via the chain 1000
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    assert np.array_equal(g.sssp(length, 0)[0], po.sssp(og, length, 0)[0])
    g.free()


def test_sssp_edge_property_follows_device_row_sort(gmx):
    """A host graph that is frozen but not semi-sorted (prepare_external_creation with caller-filled rows, e.g.
    create_uniform_random_graph_new, graph_gen.cc:12-105) is uploaded with GMX_GRAPH_SORT_ROWS: the device sorts
    the rows.  The edge property of sssp stays indexed by the CALLER's slots (the reference's sssp never sorts and
    walks the rows as stored), so the library has to carry it through the sort (e_idx2idx, gm_graph.cc:468-503)."""
    rng = np.random.default_rng(5)
    N, M = 3000, 40000
    src = rng.integers(0, N, M).astype(np.int32)
    dst = rng.integers(0, N, M).astype(np.int32)
    order = np.argsort(src, kind="stable")                # rows in arrival order: destinations unsorted, duplicates kept
    src, dst = src[order], dst[order]
    begin = np.zeros(N + 1, np.int32)
    np.cumsum(np.bincount(src, minlength=N), out=begin[1:])
    length = rng.integers(1, 50, M).astype(np.int32)
    og = po.Graph(N, begin, dst.copy())                   # the oracle walks the rows as stored
    root = int(src[0])
    want = po.sssp(og, length, root)[0]
    g = gmx.Graph.upload(begin, dst, None, None, flags=gmx.GMX_GRAPH_SORT_ROWS)
    b2, sorted_idx, _, _ = g.download()
    emap = g.edge_order()
    assert emap is not None and np.array_equal(b2, begin)
    assert np.array_equal(sorted_idx, dst[emap])           # e_idx2idx: sorted slot -> uploaded slot
    assert np.array_equal(np.sort(emap), np.arange(M))
    rows = np.repeat(np.arange(N), np.diff(begin))
    assert np.all((np.diff(sorted_idx) >= 0) | (np.diff(rows) > 0))     # every row ascending
    same = (np.diff(sorted_idx) == 0) & (np.diff(rows) == 0)
    assert np.all(np.diff(emap)[same] > 0)                 # equal destinations keep their uploaded order
    dist, _ = g.sssp(length, root)
    assert np.array_equal(dist, want)
    g.free()
    # rows already in order: no map, same answers
    g = gmx.Graph.upload(begin, sorted_idx, None, None, flags=gmx.GMX_GRAPH_SORT_ROWS)
    assert g.edge_order() is None
    assert np.array_equal(g.sssp(length[emap], root)[0], want)
    g.free()


def test_avg_teen_cnt_and_conduct(gmx, golden):
    """gmx_avg_teen_cnt / gmx_conduct against the reference-pinned fixtures and the oracle: the integer arrays
    exactly, the returned float32 bit for bit."""
    for name, c in golden["cases"].items():
        man = golden["manifest"]["rmat"].get(name) or golden["manifest"]["hand"].get(name[len("hand_"):])
        g = gmx.Graph.upload(c["begin"], c["node_idx"], c["r_begin"], c["r_node_idx"])
        for K, want in zip((5, 25, 100), man["teen_avg_K5_K25_K100"]):
            avg, cnt, _ = g.avg_teen_cnt(c["age"], K)
            assert np.array_equal(cnt, c["teen_cnt"]), name
            assert np.float32(avg).tobytes() == np.float32(want).tobytes(), (name, K, avg, want)
        for num, want in enumerate(man["conduct_0_4"]):
            got, _ = g.conduct(c["member"], num)
            assert np.float32(got).tobytes() == np.float32(want).tobytes(), (name, num, got, want)
        g.free()
    rng = np.random.default_rng(5)
    for scale, permute in [(14, False), (17, True)]:
        og = po.rmat_graph(scale, permute=permute)
        g = gmx.Graph.upload(og.begin, og.node_idx, None, None, flags=gmx.GMX_GRAPH_NO_REVERSE)
        for age in (rng.integers(0, 40, og.N).astype(np.int32), np.full(og.N, 10, np.int32), np.full(og.N, 50, np.int32)):
            for K in (5, 30):
                avg, cnt, _ = g.avg_teen_cnt(age, K)
                want_avg, want_cnt = po.avg_teen_cnt(og, age, K)
                assert np.array_equal(cnt, want_cnt)
                assert np.float32(avg).tobytes() == np.float32(want_avg).tobytes()
        for member in (rng.integers(0, 4, og.N).astype(np.int32), np.zeros(og.N, np.int32)):
            for num in (0, 1, 3, 7):
                assert np.float32(g.conduct(member, num)[0]).tobytes() == np.float32(po.conduct(og, member, num)).tobytes()
        g.free()
    # group with outgoing crossing edges but no edges counted on the smaller side: m == 0 and Cross > 0 -> FLT_MAX
    og = po.graph_from_edges(3, np.array([0], np.int32), np.array([1], np.int32))
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    member = np.array([1, 0, 0], np.int32)
    assert np.float32(g.conduct(member, 1)[0]) == np.float32(po.conduct(og, member, 1)) == np.finfo(np.float32).max
    g.free()


@pytest.mark.parametrize("per_row", [False, True])
def test_neighbour_counts_flat_and_per_row(gmx, per_row, monkeypatch):
    """avg_teen_cnt (in-neighbours, reverse CSR) and conduct (out-neighbours of the members) through the flat merge-path
    count and through round 2's one-row-per-lane count: rows cut by workgroup ranges (a star's hub: thousands of slots),
    runs of empty rows, the last rows of the graph, against the oracle."""
    if per_row:
        monkeypatch.setenv("GMX_ROWCNT_PER_ROW", "1")
    rng = np.random.default_rng(9)
    n = 5000
    star_src = np.concatenate([np.zeros(n - 1, np.int32), np.arange(1, n, dtype=np.int32)])
    star_dst = np.concatenate([np.arange(1, n, dtype=np.int32), np.zeros(n - 1, np.int32)])
    graphs = [po.rmat_graph(14, permute=False), po.rmat_graph(17, permute=True),
              po.graph_from_edges(n, star_src, star_dst),                                         # hub rows of 4999 slots
              po.graph_from_edges(9000, np.array([8999, 8999, 3], np.int32), np.array([0, 8998, 8999], np.int32))]   # almost only empty rows
    for og in graphs:
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
        for age in (rng.integers(0, 40, og.N).astype(np.int32), np.full(og.N, 15, np.int32)):
            avg, cnt, _ = g.avg_teen_cnt(age, 5)
            want_avg, want_cnt = po.avg_teen_cnt(og, age, 5)
            assert np.array_equal(cnt, want_cnt)
            assert np.float32(avg).tobytes() == np.float32(want_avg).tobytes()
        member = rng.integers(0, 3, og.N).astype(np.int32)
        for num in (0, 2, 5):
            assert np.float32(g.conduct(member, num)[0]).tobytes() == np.float32(po.conduct(og, member, num)).tobytes()
        g.free()


def test_reverse_edge_map(gmx, golden):
    """gmx_graph_reverse_edge_map = gm_graph's e_rev2idx: a one-to-one map from reverse slots to forward slots
    with swapped endpoints, copies of a repeated edge in order (what make_reverse_edges leaves after the sort)."""
    for og in (po.rmat_graph(12, permute=False), po.rmat_graph(10, permute=True)):   # RMAT keeps multi-edges
        g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
        m = g.reverse_edge_map()
        g.free()
        E = og.M
        assert np.array_equal(np.sort(m), np.arange(E))
        rdst = np.repeat(np.arange(og.N), np.diff(og.r_begin))          # destination of every reverse slot
        fsrc = np.repeat(np.arange(og.N), np.diff(og.begin))            # source of every forward slot
        assert np.array_equal(og.node_idx[m], rdst)
        assert np.array_equal(fsrc[m], og.r_node_idx)
        same = (rdst[1:] == rdst[:-1]) & (og.r_node_idx[1:] == og.r_node_idx[:-1])
        assert np.all(m[1:][same] > m[:-1][same])


@pytest.mark.parametrize("slices", [1, 2, 3, 4, 8])
def test_pagerank_slice_counts(gmx, monkeypatch, slices):
    """The sliced variant with every supported slice count (GMX_PR_SLICES; 3 does not divide the 8 XCDs, so the
    static block deal is off and everything goes through the queues), fp32 and fp64, plain and in 3 chunks."""
    monkeypatch.setenv("GMX_PR_SLICES", str(slices))
    og = po.rmat_graph(16, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    want, _, want_diff = po.pagerank(og, 1e-300, 0.85, 6)
    for elem, tol in ((4, PR_RTOL_F32), (8, PR_RTOL_F64)):
        for chunks in (1, 3):
            st = gmx.PageRankState(g, elem, 0, 1, gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED)
            st.set_chunks(chunks)
            st.reset(0.85)
            for _ in range(6):
                st.step()
            assert rel_err(st.download(), want) < tol, (slices, elem, chunks)
            assert abs(st.diff() - want_diff) <= (1e-3 if elem == 4 else 1e-9) * want_diff
            st.free()
    g.free()


def test_dist_engine_world1_and_kernel_timing(gmx):
    from dist_pagerank import DistPageRank, GmxEngine
    og = po.rmat_graph(14, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    eng = GmxEngine(gmx, g, 4, 0, 1, gmx.GMX_PR_RELABEL)
    pr = DistPageRank(eng)
    eng.state.timing(True)
    cnt, diff = pr.run(0.001, 0.85, 100)
    n, ms = eng.state.kernel_time()
    want, it, _ = po.pagerank(og, 0.001, 0.85, 100)
    assert cnt == it and n == cnt and ms > 0
    assert rel_err(eng.download(), want) < PR_RTOL_F32
    g.free()


def _check_all_kernels(gmx, og, root=0, tc=True, pr_tol=PR_RTOL_F64):
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    want, it, _ = po.pagerank(og, 0.001, 0.85, 100)
    for opts in (1, 3, 5, 7):
        st = gmx.PageRankState(g, 8, 0, 1, opts)
        st.reset(0.85)
        for _ in range(it):
            st.step()
        got = st.download()
        if og.N:
            assert rel_err(got, want) < pr_tol, (opts, rel_err(got, want))
        st.free()
    r64, s64 = g.pagerank(0.001, 0.85, 100, np.float64)
    assert s64["iterations"] == (it if og.N else 0)
    dist, _ = g.hop_dist(root)
    assert np.array_equal(dist, po.bfs_queue(og, root))
    if tc:
        assert g.triangle_counting()[0] == po.triangle_counting_merge(og)
    g.free()


def test_edge_case_hub_rows_spanning_many_blocks(gmx):
    """A star with 200 000 leaves (both directions) plus a ring: one in-row / out-row of 200 000 entries
    spans ~400 merge-path blocks and every slice; the other rows have 2-3 entries."""
    n = 200001
    leaves = np.arange(1, n, dtype=np.int32)
    src = np.concatenate([np.zeros(n - 1, np.int32), leaves, leaves])
    dst = np.concatenate([leaves, np.zeros(n - 1, np.int32), np.roll(leaves, 1)])
    og = po.graph_from_edges(n, src, dst)
    # the CPU adds the hub's 200 000 terms one by one (rounding bound n*eps = 2.2e-11), the device in a tree
    _check_all_kernels(gmx, og, root=5, pr_tol=5e-11)


def test_edge_case_mostly_empty_rows(gmx):
    """2^20 vertices, 3000 edges: long runs of empty rows between the few non-empty ones."""
    rng = np.random.default_rng(11)
    V = 1 << 20
    src = rng.integers(0, V, 3000).astype(np.int32)
    dst = rng.integers(0, V, 3000).astype(np.int32)
    og = po.graph_from_edges(V, src, dst)
    _check_all_kernels(gmx, og, root=int(src[0]))


def test_edge_case_duplicate_edges_and_self_loops(gmx):
    """Multi-edges and self loops are kept by the reference (graph_gen.cc:265-266; hand-built graphs)."""
    rng = np.random.default_rng(3)
    V = 300
    src = rng.integers(0, V, 20000).astype(np.int32)
    dst = rng.integers(0, 40, 20000).astype(np.int32)       # few destinations: many duplicates
    src = np.concatenate([src, np.arange(V, dtype=np.int32)])  # + one self loop per vertex
    dst = np.concatenate([dst, np.arange(V, dtype=np.int32)])
    og = po.graph_from_edges(V, src, dst)
    _check_all_kernels(gmx, og, root=1)


@pytest.mark.parametrize("V,E,nranks,chunks", [(100003, 1200007, 1, 1), (100003, 1200007, 3, 2), (65537, 300000, 5, 2),
                                               (4099, 20000, 2, 3), (1, 0, 1, 1), (3, 5, 2, 2)])
def test_pagerank_odd_sizes_all_variants(gmx, V, E, nranks, chunks):
    """Vertex and edge counts that are not multiples of anything (prime V, ranges padded to whole slice runs,
    a short last rank, ranks owning no row at all), every kernel variant, N ranks side by side with the sweep
    in row chunks where the variant supports it."""
    import torch
    rng = np.random.default_rng(V + E)
    src = (rng.zipf(1.4, E) % max(V, 1)).astype(np.int32) if E else np.zeros(0, np.int32)   # skewed out-degrees
    dst = rng.integers(0, max(V, 1), E).astype(np.int32)
    og = po.graph_from_edges(V, src, dst)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    iters = 6
    want, _, want_diff = po.pagerank(og, 1e-300, 0.85, iters)
    for options in (1, 3, 7):
        states = [gmx.PageRankState(g, 8, r, nranks, options) for r in range(nranks)]
        C = chunks if options == 7 else 1
        for s in states:
            s.set_chunks(C)
            s.reset(0.85)
        n = torch.as_tensor(states[0].contrib_slice(), device="cuda").numel()
        need = states[0].exchange_count()

        def exchange(bufs):
            for dst_t in bufs:
                for r, src_t in enumerate(bufs):
                    if dst_t is not src_t:
                        dst_t[r * n:r * n + need].copy_(src_t[r * n:r * n + need])
            torch.cuda.synchronize()

        exchange([torch.as_tensor(s.contrib_full(), device="cuda") for s in states])
        for _ in range(iters):
            nxt = [torch.as_tensor(s.contrib_next_full(), device="cuda") for s in states]
            for s in states:
                s.step()
            exchange(nxt)
        out = np.zeros(og.N)
        for s in states:
            s.download(out)
        if V:
            assert rel_err(out, want) < 5e-12, (options, rel_err(out, want))
        diff = sum(s.diff() for s in states)
        assert abs(diff - want_diff) <= 1e-9 * max(want_diff, 1e-30) + 1e-15
        for s in states:
            s.free()
    g.free()


def test_edge_case_empty_graphs(gmx):
    for V in (0, 1, 5):
        begin = np.zeros(V + 1, np.int32)
        g = gmx.Graph.upload(begin, np.zeros(0, np.int32), begin, np.zeros(0, np.int32))
        rank, st = g.pagerank()
        assert len(rank) == V
        if V:
            # no edges: every vertex gets (1-d)/N after the first sweep, diff = |that - 1/N| * N > e, second sweep converges
            og = po.Graph(V, begin, np.zeros(0, np.int32), begin, np.zeros(0, np.int32))
            want, it, _ = po.pagerank(og)
            assert st["iterations"] == it and np.allclose(rank, want, rtol=1e-15)
            dist, _ = g.hop_dist(0)
            assert dist[0] == 0 and (dist[1:] == INT_MAX).all()
        assert g.triangle_counting()[0] == 0
        g.free()


def test_bad_arguments_are_reported(gmx):
    with pytest.raises(gmx.GmxError):
        gmx.Graph.upload(np.array([0, 2], np.int32), np.array([0], np.int32))      # begin[V] != E
    with pytest.raises(gmx.GmxError):
        gmx.Graph.from_edges(4, np.array([0, 9], np.int32), np.array([1, 2], np.int32))   # endpoint out of range
    with pytest.raises(gmx.GmxError):
        gmx.Graph.rmat(64, 1024, a=0.6, b=0.3, c=0.2)                              # a+b+c >= 1 (graph_gen.cc:161)
    g = gmx.Graph.upload(np.array([0, 1, 1], np.int32), np.array([1], np.int32), flags=gmx.GMX_GRAPH_NO_REVERSE)
    with pytest.raises(gmx.GmxError):
        g.pagerank()                                                               # needs the reverse CSR
    g.free()


def test_pagerank_fp32_limb_guard_and_tiny_contributions(gmx, monkeypatch):
    """The fp32 binned sweep adds fp32 pair sums into ONE 2^-62 fixed-point limb; a term below 2^-39 is truncated
    (gmx_pr_cold.hip, pr_cold_limb_guard).  (1) The adversarial shape of VERDICT r2: rows whose in-neighbours have a
    huge out-degree and the minimal rank, so every contribution is (1-d)/N/outdeg = 2^-39.7 -- the ranks must still
    hold 1e-6 against the oracle.  (2) A graph and damping factor for which the guard cannot prove the bar (every
    tile holds sources of out-degree > (1-d)/N * 2^39): the stepping API refuses the fp32 plan, the whole-kernel entry
    computes with the two-limb fp64 plan and rounds once, and the result holds 1e-6."""
    # (1) 2^21 vertices (the binned sweep is the default from 2^20): 16 hubs x 65536 targets + a sparse background
    V = 1 << 21
    rng = np.random.default_rng(7)
    hubs = np.arange(16, dtype=np.int32) * 30000 + 11             # far apart in the original numbering
    targets = (1 << 20) + np.arange(1 << 16, dtype=np.int32)
    src = np.concatenate([np.repeat(hubs, len(targets)), rng.integers(1 << 19, V, 4 * V).astype(np.int32)])
    dst = np.concatenate([np.tile(targets, len(hubs)), rng.integers(1 << 19, V, 4 * V).astype(np.int32)])
    g = gmx.Graph.from_edges(V, src, dst)
    begin, node_idx, rb, rn = g.download()
    og = po.Graph(V, begin, node_idx, rb, rn)
    assert np.all(np.diff(rb)[hubs] == 0)                          # the hubs keep the teleport rank
    assert 0.15 / V / (1 << 16) < 2.0 ** -39
    want, it, _ = po.pagerank(og, 1e-300, 0.85, 5)
    st = gmx.PageRankState(g, 4, 0, 1, gmx.default_pr_options(V, 1))
    assert st.cold_info()["hot_ids"] == 0
    st.reset(0.85)
    for _ in range(5):
        st.step()
    assert rel_err(st.download(), want) < PR_RTOL_F32
    st.free()
    g.free()
    # (2) every source has out-degree 64 and d = 0.99999: (1-d)/N * 2^39 = 42 < 64 in every tile
    monkeypatch.setenv("GMX_PR_COLD", "0")
    V = 1 << 17
    src = np.repeat(np.arange(V, dtype=np.int32), 64)
    dst = rng.integers(0, V, len(src)).astype(np.int32)
    g = gmx.Graph.from_edges(V, src, dst)
    begin, node_idx, rb, rn = g.download()
    og = po.Graph(V, begin, node_idx, rb, rn)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | gmx.GMX_PR_COLD_PB
    st = gmx.PageRankState(g, 4, 0, 1, options)
    st.reset(0.85)                                                 # provable at d = 0.85 ...
    with pytest.raises(gmx.GmxError):
        st.reset(0.99999)                                          # ... not at d = 0.99999: refused, not silently imprecise
    st.free()
    st8 = gmx.PageRankState(g, 8, 0, 1, options)                   # two limbs: accepted
    st8.reset(0.99999)
    for _ in range(4):
        st8.step()
    want, _, _ = po.pagerank(og, 1e-300, 0.99999, 4)
    assert rel_err(st8.download(), want) < PR_RTOL_F64
    st8.free()
    g.free()


def test_pagerank_damping_outside_unit_interval(gmx):
    """The reference driver accepts any d > 0 (pagerank_main.cc:59-62).  With d > 1 ranks change sign and the
    contributions leave [0, 1], which the fixed-point bins cannot hold: the stepping API refuses such a d on a binned
    plan, and the whole-kernel entry runs it on the pull sweep (floating-point sums).  Compared with the oracle
    relative to the largest |rank| (ranks cross zero)."""
    V = (1 << 20) + 4096                                           # above 2^20: the default plan is the binned one
    rng = np.random.default_rng(3)
    src = rng.integers(0, V, 8 * V).astype(np.int32)
    dst = (V * rng.random(8 * V) ** 2).astype(np.int32)
    g = gmx.Graph.from_edges(V, src, dst)
    begin, node_idx, rb, rn = g.download()
    og = po.Graph(V, begin, node_idx, rb, rn)
    st = gmx.PageRankState(g, 8, 0, 1, gmx.default_pr_options(V, 1))
    assert st.cold_info()["hot_ids"] == 0
    for bad in (1.5, 0.0, -0.2, float("nan")):
        with pytest.raises(gmx.GmxError):
            st.reset(bad)
    st.reset(1.0)
    st.free()
    for d in (1.5, 1.0):
        want, it, _ = po.pagerank(og, 1e-300, d, 6)
        for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-6)):
            rank, stt = g.pagerank(1e-300, d, 6, dt)
            assert stt["iterations"] == it
            assert float(np.max(np.abs(rank.astype(np.float64) - want))) < tol * float(np.max(np.abs(want))), (d, dt)
    g.free()


@pytest.mark.parametrize("nranks", [2, 3])
def test_bench_multi_rank_rehearsal(nranks):
    """bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), with the ranks sharing
    this box's one GPU (GMX_BENCH_SHARED_GPU=1: gloo + host barrier instead of RCCL; the peer pushes go through hipIpc as
    on a real node).  What it proves is this file's N > 1 code path end to end: warm-up agreement, the pipelined pushed
    step, the replica check after the timed steps (no fallback), one JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, GMX_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--scale", "22",
                        "--steps", "4", "--warmup", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == nranks and out["steps"] == 4 and out["value"] > 0
    cfg = out["config"]
    assert cfg["exchange_check"] == "replicas equal an all-gather of the owned slices on every rank"
    assert cfg["step_form"] == "pushed, pipelined", cfg["step_form"]      # no fallback
    assert cfg["exchange_packed"] is True and cfg["exchange_bytes_per_rank_and_step"] > 0
    assert "rehearsal" in cfg
