"""CPU: the oracle restatement against the committed reference-generated fixtures
(tests/golden/, produced by oracle/make_golden.py from the compiled reference runtime)."""
import hashlib
import os

import numpy as np
import pytest

import pyoracle as po

INT_MAX = 2147483647


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_drand48_known_values():
    # glibc drand48 after srand48(1997), cross-checked against the reference build in make_golden.py;
    # the LCG constants are the SVID ones.
    s = po.drand48_stream(0, 3)
    # srand48(0): X0 = 0x330E; X1 = (0x5DEECE66D*0x330E + 0xB) mod 2^48
    x1 = (0x5DEECE66D * 0x330E + 0xB) & ((1 << 48) - 1)
    assert s[0] == x1 / 2.0 ** 48
    assert abs(s[0] - 0.17082803610628972) < 1e-15   # well-known first drand48() value for seed 0


def check_counts(g, c, m):
    """avg_teen_cnt / conduct of the oracle against the reference-pinned values (floats compared by value of
    the float32; FLT_MAX for an empty group with crossing edges cannot occur, 0 for an empty group can)."""
    for K, want in zip((5, 25, 100), m["teen_avg_K5_K25_K100"]):
        avg, cnt = po.avg_teen_cnt(g, c["age"], K)
        assert np.float32(avg) == np.float32(want)
        assert np.array_equal(cnt, c["teen_cnt"])
    for num, want in enumerate(m["conduct_0_4"]):
        assert np.float32(po.conduct(g, c["member"], num)) == np.float32(want)


def same_f32(a, b):
    """float32 arrays equal bit for bit, any NaN counting as NaN (its sign / payload is the platform's)."""
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))


def check_common_nbrs(g, c, m):
    """gm_common_neighbor_iter: counts on the fixture's pairs (pinned item for item against the compiled reference
    class when the fixtures were made), and triangle counting written with the iterator."""
    if "cn_src" in c:
        got = [len(po.common_nbrs(g, s, d)) for s, d in zip(c["cn_src"], c["cn_dst"])]
        assert got == c["cn_counts"].tolist()
        for s, d in list(zip(c["cn_src"], c["cn_dst"]))[:40]:        # the definition: Foreach(u: s.Nbrs)(d.isNbr(u))
            row_s = g.node_idx[g.begin[s]:g.begin[s + 1]]
            row_d = g.node_idx[g.begin[d]:g.begin[d + 1]]
            assert np.array_equal(po.common_nbrs(g, s, d), row_s[np.isin(row_s, row_d)])
    if m.get("tc_cn") is not None:
        assert po.triangle_counting_cn(g) == m["tc_cn"]


def check_bc(g, c):
    """comp_BC (bc.gm): this fork's form and upstream's, against the reference-pinned arrays."""
    assert same_f32(po.bc(g, c["bc_seeds"], False), c["bc"])
    assert same_f32(po.bc(g, c["bc_seeds"], True), c["bc_skip_root"])
    assert not np.isnan(c["bc_skip_root"]).any()


@pytest.mark.parametrize("name", ["rmat6_noperm", "rmat6_perm", "rmat8_noperm", "rmat8_perm",
                                  "rmat10_noperm", "rmat10_perm"])
def test_rmat_fixture_full(golden, name):
    c = golden["cases"][name]
    m = golden["manifest"]["rmat"][name]
    begin, raw, att = po.rmat_raw_csr(m["N"], m["M"], m["seed"], *m["abc"], permute=m["permute"])
    assert att == m["attempts"]
    assert np.array_equal(begin, c["begin"]) and np.array_equal(raw, c["raw_node_idx"])
    g = po.Graph(m["N"], begin, raw).prepare()
    assert np.array_equal(g.node_idx, c["node_idx"])
    assert np.array_equal(g.r_begin, c["r_begin"]) and np.array_equal(g.r_node_idx, c["r_node_idx"])
    rank, it, _ = po.pagerank(g, 0.001, 0.85, 100, nthreads=1)
    assert it == m["pr_iters"] and np.array_equal(rank, c["rank"])
    rank20, it20, _ = po.pagerank(g, 1e-300, 0.85, 20, nthreads=4)
    assert it20 == 20 and np.array_equal(rank20, c["rank20"])
    dist, _ = po.hop_dist(g, m["root"])
    assert np.array_equal(dist, c["dist"])
    assert np.array_equal(po.bfs_queue(g, m["root"]), c["dist"])
    assert np.array_equal(po.sssp(g, c["sssp_len"], m["root"])[0], c["sssp_dist"])
    assert sha(c["sssp_len"]) == m["sha_sssp_len"] and sha(c["sssp_dist"]) == m["sha_sssp_dist"]
    check_counts(g, c, m)
    check_bc(g, c)
    check_common_nbrs(g, c, m)
    assert po.triangle_counting(g) == m["tc_directed"]
    assert po.triangle_counting_merge(g) == m["tc_directed"]
    gs = po.symmetrize(g)
    assert gs.M == m["M_sym"]
    assert po.triangle_counting_merge(gs) == m["tc_symmetrized"]


@pytest.mark.parametrize("name", ["rmat12_noperm", "rmat14_noperm", "rmat14_perm", "rmat16_noperm"])
def test_rmat_fixture_hashed(golden, name):
    """Bigger graphs are pinned by sha256 of the reference outputs (manifest.json)."""
    m = golden["manifest"]["rmat"][name]
    begin, raw, att = po.rmat_raw_csr(m["N"], m["M"], m["seed"], *m["abc"], permute=m["permute"])
    assert att == m["attempts"]
    assert sha(begin) == m["sha_begin"] and sha(raw) == m["sha_raw_node_idx"]
    g = po.Graph(m["N"], begin, raw).prepare()
    assert sha(g.node_idx) == m["sha_node_idx"]
    assert sha(g.r_begin) == m["sha_r_begin"] and sha(g.r_node_idx) == m["sha_r_node_idx"]
    rank, it, _ = po.pagerank(g, 0.001, 0.85, 100, nthreads=1)
    assert it == m["pr_iters"] and sha(rank) == m["sha_rank_f64"]
    dist, _ = po.hop_dist(g, m["root"])
    assert sha(dist) == m["sha_dist"]
    assert int((dist != INT_MAX).sum()) == m["reached"]
    if "sha_bc" in m:
        seeds = np.array(m["bc_seeds"], np.int32)
        for skip, key in ((False, "bc"), (True, "bc_skip_root")):
            got = po.bc(g, seeds, skip)
            got[np.isnan(got)] = np.float32(np.nan)
            assert sha(got) == m["sha_" + key] and int(np.isnan(got).sum()) == m[key + "_nan"]
    if m["tc_directed"] is not None:
        assert po.triangle_counting_merge(g) == m["tc_directed"]
    if m.get("tc_cn") is not None:
        assert po.triangle_counting_cn(g) == m["tc_cn"]
    gsym = po.symmetrize(g)
    assert po.triangle_counting_merge(gsym) == m["tc_symmetrized"]
    if m["N"] <= (1 << 14):
        assert po.triangle_counting_cn(gsym) == m["tc_symmetrized"]     # symmetric graph: both forms count triangles


def test_hand_graphs(golden):
    for name, m in golden["manifest"]["hand"].items():
        c = golden["cases"]["hand_" + name]
        g = po.Graph(m["N"], c["begin"].copy(), c["raw_node_idx"].copy()).prepare()
        assert np.array_equal(g.node_idx, c["node_idx"]), name
        assert np.array_equal(g.r_node_idx, c["r_node_idx"]), name
        rank, it, _ = po.pagerank(g, nthreads=1)
        assert it == m["pr_iters"] and np.array_equal(rank, c["rank"]), name
        dist, _ = po.hop_dist(g, m["root"])
        assert np.array_equal(dist, c["dist"]), name
        assert np.array_equal(po.sssp(g, c["sssp_len"], m["root"])[0], c["sssp_dist"]), name
        check_counts(g, c, m)
        check_bc(g, c)
        check_common_nbrs(g, c, m)
        assert po.triangle_counting(g) == m["tc"], name
        assert po.triangle_counting_merge(g) == m["tc"], name


def test_tutorial_indegree(golden):
    # doc/tutorial.md:241-279 -- the in-degree sum over the 5-node example is the edge count
    c = golden["cases"]["hand_tutorial5"]
    assert int(np.diff(c["r_begin"]).sum()) == len(c["node_idx"])


def test_binary_format_fixture(golden, tmp_path):
    """The .bin written by the REFERENCE's store_binary loads in the oracle, and the oracle
    writes the same bytes (gm_graph_binary_loader.cc:19-40,207-252)."""
    from conftest import GOLD
    m = golden["manifest"]["bin"]
    path = os.path.join(GOLD, m["file"])
    data = open(path, "rb").read()
    assert hashlib.sha256(data).hexdigest() == m["sha256"]
    g = po.load_binary(path)
    assert (g.N, g.M) == (m["N"], m["M"])
    out = str(tmp_path / "o.bin")
    po.store_binary(out, g)
    assert open(out, "rb").read() == data
    c = golden["cases"]["rmat8_noperm"]
    assert np.array_equal(g.begin, c["begin"]) and np.array_equal(g.node_idx, c["node_idx"])
    assert np.array_equal(g.r_node_idx, c["r_node_idx"])


def test_hop_dist_properties():
    g = po.rmat_graph(12, permute=True, seed=7)
    root = int(np.argmax(np.diff(g.begin)))
    dist, levels = po.hop_dist(g, root)
    # every edge (n -> s) with n reached satisfies dist[s] <= dist[n] + 1
    src = np.repeat(np.arange(g.N), np.diff(g.begin))
    r = dist[src] != INT_MAX
    assert (dist[g.node_idx[r]] <= dist[src[r]] + 1).all()
    assert levels == dist[dist != INT_MAX].max() + 1
