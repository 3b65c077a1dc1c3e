#!/usr/bin/env python3
"""Pin the CPU oracle against the compiled reference runtime and (re)generate
the committed fixtures under tests/golden/.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference
to build oracle/_ref/libgmref.so via `make -C oracle ref`).  Nothing here is
imported by the product, the GPU tests, smoke() or bench.py; those read the
.npz / .json / .bin fixtures this script wrote.

  python oracle/make_golden.py            # pin + rewrite fixtures
  python oracle/make_golden.py --check    # pin only, do not rewrite

What is pinned (oracle function  <-  reference code actually executed):
  gmo_drand48                 <- glibc srand48/drand48 (what graph_gen.cc calls)
  gmo_create_rmat_graph       <- create_RMAT_graph            (graph_gen.cc:159-287)
  gmo_semi_sort / gmo_make_reverse_edges
                              <- gm_graph::do_semi_sort / make_reverse_edges (gm_graph.cc)
  gmo_get_edge_idx_for_src_dest <- gm_graph::get_edge_idx_for_src_dest (gm_graph.cc:589-633)
  gmo_store_binary/load_binary <- gm_graph::store_binary / load_binary
  gmo_pagerank                <- plain emission run on the reference runtime (ref_harness.cc)
  gmo_hop_dist, gmo_bfs_queue <- emission on the reference runtime AND gm_bfs_template levels
  gmo_triangle_counting(+merge) <- emission with gm_graph::is_neighbor
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import pyoracle as po  # noqa: E402

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
INT_MAX = 2147483647


def ref_lib():
    subprocess.check_call(["make", "-C", HERE, "-j8", "ref"], stdout=subprocess.DEVNULL)
    R = C.CDLL(os.path.join(HERE, "_ref", "libgmref.so"))
    R.ref_drand48_stream.argtypes = [C.c_long, C.c_int, f64p]
    R.ref_rmat_csr.argtypes = [C.c_int32, C.c_int32, C.c_long, C.c_double, C.c_double, C.c_double, C.c_int, i32p, i32p]
    R.ref_prepare.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p, i32p]
    R.ref_store_binary.argtypes = [C.c_char_p, C.c_int32, C.c_int32, i32p, i32p]
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    R.ref_common_nbrs.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int32, i32p, i32p, i64p, i32p, C.c_int64]
    R.ref_common_nbrs.restype = C.c_int64
    R.ref_triangle_counting_cn.argtypes = [C.c_int32, C.c_int32, i32p, i32p]
    R.ref_triangle_counting_cn.restype = C.c_int64
    R.ref_uniform_graph.argtypes = [C.c_int32, C.c_int32, C.c_long, C.c_int, i32p, i32p]
    R.ref_rand32_min.argtypes = [C.c_long, C.c_int32]
    R.ref_rand32_min.restype = C.c_int32
    R.ref_bc.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int32, C.c_int, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.c_int]
    R.ref_load_adj.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32p, i32p, C.c_int32, C.c_int32]
    R.ref_load_binary.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_pagerank.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_double, C.c_double, C.c_int32, f64p,
                               C.c_int, C.POINTER(C.c_int32)]
    R.ref_hop_dist.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int32, i32p, C.c_int]
    R.ref_sssp.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int32, i32p, C.c_int]
    R.ref_avg_teen_cnt.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p, C.c_int32, C.c_int]
    R.ref_avg_teen_cnt.restype = C.c_float
    R.ref_conduct.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int32, C.c_int]
    R.ref_conduct.restype = C.c_float
    R.ref_bfs_levels.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int32, i32p, C.c_int, C.c_int]
    R.ref_triangle_counting.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int]
    R.ref_triangle_counting.restype = C.c_int64
    R.ref_is_neighbor_many.argtypes = [C.c_int32, C.c_int32, i32p, i32p, C.c_int, i32p, i32p, i32p]
    return R


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def e32(n):
    return np.empty(max(n, 1), np.int32)[:n].copy()


def ref_graph(R, N, M, seed, a, b, c, permute):
    begin = np.empty(N + 1, np.int32)
    raw = e32(M)
    R.ref_rmat_csr(N, M, seed, a, b, c, int(permute), begin, raw)
    snode = e32(M)
    rb = np.empty(N + 1, np.int32)
    rn = e32(M)
    R.ref_prepare(N, M, begin, raw, snode, rb, rn)
    return begin, raw, snode, rb, rn


def ref_kernels(R, N, begin, node_idx, root=0, pr_args=(0.001, 0.85, 100), tc=True):
    M = len(node_idx)
    rank = np.empty(max(N, 1), np.float64)[:N].copy()
    it = C.c_int32(0)
    R.ref_pagerank(N, M, begin, node_idx, pr_args[0], pr_args[1], pr_args[2], rank, 1, C.byref(it))
    dist = e32(N)
    R.ref_hop_dist(N, M, begin, node_idx, root, dist, 4)
    lv = e32(N)
    R.ref_bfs_levels(N, M, begin, node_idx, root, lv, 4, 1)
    lv2 = e32(N)
    R.ref_bfs_levels(N, M, begin, node_idx, root, lv2, 1, 0)
    assert np.array_equal(lv, lv2), "gm_bfs_template: forward-only vs reverse-assisted differ"
    assert np.array_equal(dist, lv), "emitted hop_dist vs gm_bfs_template levels differ"
    T = int(R.ref_triangle_counting(N, M, begin, node_idx, 4)) if tc else None
    return rank, it.value, dist, T


def sssp_lengths(M, salt):
    """Edge lengths 1..100 as sssp_main.cc:33-34 draws them (there from gm_rand32; any positive lengths do)."""
    return np.random.default_rng(1000003 * salt + M).integers(1, 101, M).astype(np.int32)


def check_sssp(R, name, g, root, salt):
    """sssp on the semi-sorted graph: oracle restatement == emission on the reference runtime == scipy Dijkstra."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph as csg
    length = sssp_lengths(g.M, salt)
    dist_r = e32(g.N)
    R.ref_sssp(g.N, g.M, g.begin, g.node_idx, length, root, dist_r, 4)
    dist_o, _ = po.sssp(g, length, root, nthreads=8)
    assert np.array_equal(dist_o, dist_r), (name, "sssp")
    dist_1, _ = po.sssp(g, length, root, nthreads=1)
    assert np.array_equal(dist_1, dist_r), (name, "sssp 1 thread")
    if g.M and 0 <= root < g.N:
        src = np.repeat(np.arange(g.N), np.diff(g.begin))
        order = np.lexsort((length, g.node_idx, src))            # cheapest copy of every parallel edge first
        s, t, w = src[order], g.node_idx[order], length[order]
        first = np.ones(len(s), bool)
        first[1:] = (s[1:] != s[:-1]) | (t[1:] != t[:-1])
        A = sp.csr_matrix((w[first].astype(np.float64), (s[first], t[first])), shape=(g.N, g.N))
        dd = csg.dijkstra(A, indices=root)
        want = np.where(np.isinf(dd), INT_MAX, dd).astype(np.int64)
        assert np.array_equal(dist_r.astype(np.int64), want), (name, "sssp vs scipy")
    return length, dist_r


def node_props(N, salt):
    """Node properties for avg_teen_cnt (ages 0..39) and conduct (4 groups 10/20/30/40 % as conduct_main.cc:30-38)."""
    rng = np.random.default_rng(7919 * salt + N)
    age = rng.integers(0, 40, max(N, 1)).astype(np.int32)[:N]
    r = rng.integers(0, 100, max(N, 1))[:N]
    member = np.where(r < 10, 0, np.where(r < 30, 1, np.where(r < 60, 2, 3))).astype(np.int32)
    return age, member


def check_counts(R, name, g, salt):
    """avg_teen_cnt and conduct: oracle restatement == emission on the reference runtime (integers and the
    final float bit for bit), for K in {5, 25, 100} and every group."""
    age, member = node_props(g.N, salt)
    out = {"age": age, "member": member}
    avgs = []
    for K in (5, 25, 100):
        cnt_r = e32(g.N)
        avg_r = R.ref_avg_teen_cnt(g.N, g.M, g.begin, g.node_idx, age, cnt_r, K, 4)
        avg_o, cnt_o = po.avg_teen_cnt(g, age, K, nthreads=8)
        assert np.array_equal(cnt_o, cnt_r), (name, "teen_cnt")
        assert np.float32(avg_r).tobytes() == np.float32(avg_o).tobytes(), (name, "avg", K, avg_r, avg_o)
        avgs.append(float(np.float32(avg_r)))
        out["teen_cnt"] = cnt_r
    cs = []
    for num in range(5):   # group 4 is empty: exercises the m == 0 branch
        c_r = R.ref_conduct(g.N, g.M, g.begin, g.node_idx, member, num, 4)
        c_o = po.conduct(g, member, num, nthreads=8)
        assert np.float32(c_r).tobytes() == np.float32(c_o).tobytes(), (name, "conduct", num, c_r, c_o)
        cs.append(float(np.float32(c_r)))
    return out, avgs, cs


def check_common_nbrs(R, name, g, salt, tc=True):
    """gm_common_neighbor_iter: the oracle restatement against the compiled reference class, item for item, on edge
    pairs (s, d = a neighbour of s), reversed pairs and random pairs; and triangle counting written with it."""
    rng = np.random.default_rng(1000 + salt)
    if g.N == 0:
        return None
    rows = np.repeat(np.arange(g.N, dtype=np.int32), np.diff(g.begin))
    pick = rng.integers(0, max(g.M, 1), 200) if g.M else np.zeros(0, np.int64)
    src = np.concatenate([rows[pick], g.node_idx[pick], rng.integers(0, g.N, 100).astype(np.int32)]).astype(np.int32)
    dst = np.concatenate([g.node_idx[pick], rows[pick], rng.integers(0, g.N, 100).astype(np.int32)]).astype(np.int32)
    counts = np.zeros(len(src), np.int64)
    cap = int(np.diff(g.begin)[src].sum()) + 1
    items = np.zeros(cap, np.int32)
    total = R.ref_common_nbrs(g.N, g.M, g.begin, g.node_idx, len(src), src, dst, counts, items, cap)
    assert total <= cap
    at = 0
    for i in range(len(src)):
        got = po.common_nbrs(g, src[i], dst[i])
        assert len(got) == counts[i] and np.array_equal(got, items[at:at + counts[i]]), (name, "common_nbrs", int(src[i]), int(dst[i]))
        at += int(counts[i])
    T = None
    if tc:
        T = int(R.ref_triangle_counting_cn(g.N, g.M, g.begin, g.node_idx))
        assert po.triangle_counting_cn(g) == T, (name, "tc_cn")
    return {"cn_src": src, "cn_dst": dst, "cn_counts": counts, "tc_cn": T}


def bc_seeds(N, root):
    """Five seeds like the driver's (bc_main.cc:34-41 draws 5), fixed here: root, a few spread ones, the last vertex."""
    return np.array([root % N, (root + N // 3) % N, (7 * root + 5) % N, N - 1, N // 2], np.int32)


def canon_nan(a):
    """NaN payload / sign is the platform's (x86 SSE: the negative default NaN); compare NaN as NaN."""
    a = a.copy()
    a[np.isnan(a)] = np.float32(np.nan)
    return a


def brandes_f64(g, seeds):
    """Second opinion for comp_BC's upstream form, independent of both the oracle and the reference: Brandes'
    dependency accumulation with path counts per edge SLOT (repeated edges count repeatedly, as both emitted loops
    do), plain Python, float64."""
    N = g.N
    bc = np.zeros(N, np.float64)
    begin, idx = g.begin, g.node_idx
    for s in seeds:
        level = np.full(N, -2, np.int64)
        sigma = np.zeros(N, np.float64)
        delta = np.zeros(N, np.float64)
        level[s], sigma[s] = 0, 1.0
        order, frontier = [int(s)], [int(s)]
        while frontier:
            nxt = []
            for v in frontier:
                for u in idx[begin[v]:begin[v + 1]]:
                    if level[u] == -2:
                        level[u] = level[v] + 1
                        nxt.append(int(u))
            for v in frontier:                      # every slot into the next level carries v's path count
                for u in idx[begin[v]:begin[v + 1]]:
                    if level[u] == level[v] + 1:
                        sigma[u] += sigma[v]
            order += nxt
            frontier = nxt
        for v in reversed(order):
            if v == s:
                continue
            acc = 0.0
            for u in idx[begin[v]:begin[v + 1]]:
                if level[u] == level[v] + 1:
                    acc += sigma[v] / sigma[u] * (1.0 + delta[u])
            delta[v] = acc
            bc[v] += acc
    return bc


def check_bc(R, name, g, root):
    """comp_BC: the oracle restatement against (1) the emitted visit bodies on the reference's gm_bfs_template
    (traversal, levels, is_down_edge and both sweeps are pure reference code), run single-threaded exactly as
    emitted, and (2) an independent float64 Brandes for the upstream form.  (1) must agree bit for bit unless the
    template went bottom-up (ST_RD -> ST_RRD, gm_bfs_template.h:400-403: taken even with save_child), where it
    records no down edges and its BC is too small; those cases are listed in the manifest and rest on (2)."""
    seeds = bc_seeds(g.N, root)
    out = {"bc_seeds": seeds}
    for skip, key in ((0, "bc"), (1, "bc_skip_root")):
        # one thread only: with several, iterate_neighbor_que sets a child's visited bit before it stores the child's
        # level (gm_bfs_template.h:596-620), and a second parent that runs in between finds the bit set, reads a
        # stale level and does not record its down edge -- the reference's multi-threaded DownNbrs (hence BC) varies
        # from run to run (seen on rmat10_perm).  The single-threaded run is the deterministic definition.
        emitted = np.zeros(g.N, np.float32)
        assert R.ref_bc(g.N, g.M, g.begin, g.node_idx, seeds, len(seeds), skip, emitted, 1) == 0
        got = po.bc(g, seeds, bool(skip))
        same = canon_nan(got).tobytes() == canon_nan(emitted).tobytes()
        out[key + "_reference_equal"] = bool(same)
        if not same:
            # never more than the definition gives: the reference only loses down edges
            ok = ~np.isnan(got) & ~np.isnan(emitted)
            assert np.all(emitted[ok] <= got[ok] * (1 + 1e-5) + 1e-30), (name, key, "reference larger than the definition")
            print("  %s %s: the reference's template went bottom-up and dropped down edges (%d of %d values differ)"
                  % (name, key, int(np.sum(canon_nan(got) != canon_nan(emitted)) - np.sum(np.isnan(got) & np.isnan(emitted))), g.N))
        if skip and g.N <= (1 << 12):
            ref64 = brandes_f64(g, seeds)
            assert not np.isnan(got).any(), (name, key, "NaN in the upstream form")
            err = np.abs(got.astype(np.float64) - ref64) / np.maximum(np.abs(ref64), 1e-30)
            assert float(err[ref64 > 0].max(initial=0.0)) < 2e-4 and np.all(got[ref64 == 0] == 0), (name, key, "Brandes", float(err.max()))
        out[key] = canon_nan(got)
    return out


def bc_arrays(bcs):
    return {k: v for k, v in bcs.items() if isinstance(v, np.ndarray)}


def check_oracle_on(R, name, N, begin, raw_or_sorted, pr_args=(0.001, 0.85, 100), root=0, tc=True):
    """Run reference + oracle on one CSR; assert agreement; return results."""
    g = po.Graph(N, begin.copy(), raw_or_sorted.copy()).prepare()
    rank_r, it_r, dist_r, T_r = ref_kernels(R, N, begin, raw_or_sorted, root, pr_args, tc)
    rank_o, it_o, _ = po.pagerank(g, *pr_args, nthreads=1)
    assert it_o == it_r, (name, it_o, it_r)
    assert np.array_equal(rank_o, rank_r), (name, "pagerank 1-thread not bit-identical",
                                            np.abs(rank_o - rank_r).max())
    rank_o8, it_o8, _ = po.pagerank(g, *pr_args, nthreads=8)
    assert it_o8 == it_r and np.array_equal(rank_o8, rank_r), (name, "pagerank 8-thread")
    dist_o, _ = po.hop_dist(g, root, nthreads=8)
    assert np.array_equal(dist_o, dist_r), (name, "hop_dist")
    assert np.array_equal(po.bfs_queue(g, root), dist_r), (name, "bfs_queue")
    if tc:
        assert po.triangle_counting(g) == T_r, (name, "tc")
        assert po.triangle_counting_merge(g) == T_r, (name, "tc merge", po.triangle_counting_merge(g), T_r)
    return g, rank_r, it_r, dist_r, T_r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    R = ref_lib()
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"generator": "oracle/make_golden.py", "reference": "libshoal/Green-Marl @ /root/reference",
                "rmat": {}, "hand": {}}

    # ---- 1. drand48 ----
    ref_s = np.empty(4096, np.float64)
    R.ref_drand48_stream(1997, 4096, ref_s)
    assert np.array_equal(po.drand48_stream(1997, 4096), ref_s), "drand48 LCG mismatch"
    for seed in (0, 1, 12345, -7, 2 ** 31 + 5):
        t = np.empty(64, np.float64)
        R.ref_drand48_stream(seed, 64, t)
        assert np.array_equal(po.drand48_stream(seed, 64), t), seed
    print("drand48: pinned")

    # ---- 2. RMAT graphs + kernels ----
    fixtures = {}
    cases = [(6, False), (6, True), (8, False), (8, True), (10, False), (10, True), (12, False),
             (14, False), (14, True), (16, False)]
    for scale, perm in cases:
        N, M = 1 << scale, 16 << scale
        name = "rmat%d_%s" % (scale, "perm" if perm else "noperm")
        begin, raw, snode, rb, rn = ref_graph(R, N, M, 1997, 0.57, 0.19, 0.19, perm)
        ob, oraw, att = po.rmat_raw_csr(N, M, 1997, 0.57, 0.19, 0.19, perm)
        assert np.array_equal(ob, begin) and np.array_equal(oraw, raw), (name, "rmat csr")
        g = po.Graph(N, ob, oraw).prepare()
        assert np.array_equal(g.node_idx, snode), (name, "semi sort")
        assert np.array_equal(g.r_begin, rb) and np.array_equal(g.r_node_idx, rn), (name, "reverse")
        root = 0
        if perm:  # root 0 may be isolated on permuted graphs: pick the max-out-degree vertex, recorded
            root = int(np.argmax(np.diff(begin)))
        tc = scale <= 14
        _, rank, it, dist, T = check_oracle_on(R, name, N, begin, raw, root=root, tc=tc)
        # tight-tolerance fixed-iteration run (used by the throughput bench)
        _, rank20, it20, _, _ = check_oracle_on(R, name + "_20it", N, begin, raw,
                                                pr_args=(1e-300, 0.85, 20), root=root, tc=False)
        assert it20 == 20
        sssp_len, sssp_dist = check_sssp(R, name, g, root, scale * 2 + perm)
        props, teen_avgs, conducts = check_counts(R, name, g, scale * 2 + perm)
        # symmetrised graph for the TC measurement config
        gs = po.symmetrize(g)
        Ts = None
        if scale <= 12:
            Ts = int(R.ref_triangle_counting(gs.N, gs.M, gs.begin, gs.node_idx, 4))
            assert po.triangle_counting(gs) == Ts and po.triangle_counting_merge(gs) == Ts, name
        else:
            Ts = po.triangle_counting_merge(gs)
        # is_neighbor probes
        rng = np.random.default_rng(scale * 2 + perm)
        qs = rng.integers(0, N, 2000).astype(np.int32)
        qt = rng.integers(0, N, 2000).astype(np.int32)
        # half of the probes are real edges
        eidx = rng.integers(0, M, 1000)
        src_of = np.repeat(np.arange(N, dtype=np.int32), np.diff(begin))
        qs[:1000] = src_of[eidx]
        qt[:1000] = snode[eidx]
        out_r = e32(2000)
        R.ref_is_neighbor_many(N, M, begin, raw, 2000, qs, qt, out_r)
        L = po.lib()
        out_o = np.array([L.gmo_get_edge_idx_for_src_dest(g.begin, g.node_idx, int(s), int(t))
                          for s, t in zip(qs, qt)], np.int32)
        assert np.array_equal(out_o, out_r), (name, "is_neighbor")

        bcs = check_bc(R, name, g, root) if scale <= 14 else None
        cn = check_common_nbrs(R, name, g, scale * 2 + perm, tc=scale <= 14)
        entry = {"N": N, "M": M, "seed": 1997, "abc": [0.57, 0.19, 0.19], "permute": perm, "attempts": att,
                 "root": root, "pr_iters": it, "tc_directed": T, "tc_symmetrized": Ts, "M_sym": gs.M,
                 "reached": int((dist != INT_MAX).sum()), "max_level": int(dist[dist != INT_MAX].max()),
                 "sha_begin": sha(begin), "sha_raw_node_idx": sha(raw), "sha_node_idx": sha(snode),
                 "sha_r_begin": sha(rb), "sha_r_node_idx": sha(rn), "sha_dist": sha(dist),
                 "sha_rank_f64": sha(rank), "sha_rank20_f64": sha(rank20),
                 "props_salt": scale * 2 + perm, "teen_avg_K5_K25_K100": teen_avgs, "conduct_0_4": conducts,
                 "sha_teen_cnt": sha(props["teen_cnt"]), "sha_age": sha(props["age"]), "sha_member": sha(props["member"]),
                 "sssp_salt": scale * 2 + perm, "sha_sssp_len": sha(sssp_len), "sha_sssp_dist": sha(sssp_dist),
                 "sssp_reached": int((sssp_dist != INT_MAX).sum()), "sssp_max": int(sssp_dist[sssp_dist != INT_MAX].max()),
                 "rank_sum": float(rank.sum()), "rank_head": [float(x) for x in rank[:4]]}
        entry.update({"tc_cn": cn["tc_cn"], "sha_cn_src": sha(cn["cn_src"]), "sha_cn_dst": sha(cn["cn_dst"]), "sha_cn_counts": sha(cn["cn_counts"]),
                      "cn_salt": scale * 2 + perm})
        if bcs is not None:
            entry.update({"bc_seeds": [int(x) for x in bcs["bc_seeds"]], "sha_bc": sha(bcs["bc"]), "sha_bc_skip_root": sha(bcs["bc_skip_root"]),
                          "bc_nan": int(np.isnan(bcs["bc"]).sum()), "bc_skip_root_nan": int(np.isnan(bcs["bc_skip_root"]).sum()),
                          "bc_reference_equal": [bcs["bc_reference_equal"], bcs["bc_skip_root_reference_equal"]]})
        manifest["rmat"][name] = entry
        if scale <= 10:   # small enough to commit in full
            fixtures[name] = dict(begin=begin, raw_node_idx=raw, node_idx=snode, r_begin=rb, r_node_idx=rn,
                                  rank=rank, rank20=rank20, dist=dist, sssp_len=sssp_len, sssp_dist=sssp_dist,
                                  age=props["age"], member=props["member"], teen_cnt=props["teen_cnt"],
                                  cn_src=cn["cn_src"], cn_dst=cn["cn_dst"], cn_counts=cn["cn_counts"], **bc_arrays(bcs))
        print("%s: pinned (iters=%d, reached=%d, T=%s, Tsym=%s)" % (name, it, entry["reached"], T, Ts))

    # ---- 3. hand graphs ----
    hand = {}
    # doc/tutorial.md:241-279 five-node example (expected in-degree sums 5 4 4 0 4 ... total 17)
    tut_edges = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 4), (2, 3), (2, 4), (3, 4), (4, 0)]
    hand["empty1"] = (1, [], 0)
    hand["two_isolated"] = (2, [], 1)
    hand["self_loop"] = (3, [(0, 0), (0, 1), (1, 1), (1, 2), (2, 0)], 0)
    hand["multi_edge"] = (4, [(0, 1), (0, 1), (0, 2), (1, 2), (1, 2), (2, 0), (2, 1), (1, 0), (0, 3), (3, 0), (3, 1), (1, 3)], 0)
    hand["tutorial5"] = (5, tut_edges, 0)
    hand["path8"] = (8, [(i, i + 1) for i in range(7)], 0)
    hand["path8_mid"] = (8, [(i, i + 1) for i in range(7)], 3)
    hand["star64"] = (64, [(0, i) for i in range(1, 64)] + [(i, 0) for i in range(1, 64)], 0)
    k6 = [(i, j) for i in range(6) for j in range(6) if i != j]
    hand["k6"] = (6, k6, 2)
    # the reference's own sample adjacency lists, read by ITS loader (sparse keys renumbered, rows semi-sorted)
    for nm, rel in (("adj_very_small_sample", "apps/output_cpp/gm_graph/test/very_small_sample.adj"),
                    ("adj_cpp_be_test", "test/cpp_be/test.adj")):
        N_, M_ = C.c_int32(0), C.c_int32(0)
        ab, an = np.zeros(4097, np.int32), np.zeros(65536, np.int32)
        rc = R.ref_load_adj(os.path.join("/root/reference", rel).encode(), C.byref(N_), C.byref(M_), ab,
                            an, 4096, 65536)
        assert rc == 0, (nm, rc)
        ab, an = ab[:N_.value + 1], an[:M_.value]
        hand[nm] = (N_.value, [(int(v), int(an[e])) for v in range(N_.value) for e in range(ab[v], ab[v + 1])], 0)
        print("%s: %d vertices, %d edges from the reference's adjacency loader" % (nm, N_.value, M_.value))
    for name, (N, edges, root) in hand.items():
        src = np.array([e[0] for e in edges], np.int32)
        dst = np.array([e[1] for e in edges], np.int32)
        begin, raw = po.csr_from_edges(N, src, dst)
        g, rank, it, dist, T = check_oracle_on(R, name, N, begin, raw, root=root, tc=True)
        sssp_len, sssp_dist = check_sssp(R, name, g, root, len(name))
        props, teen_avgs, conducts = check_counts(R, name, g, len(name))
        bcs = check_bc(R, name, g, root)
        cn = check_common_nbrs(R, name, g, len(name)) if len(edges) else None
        fixtures["hand_" + name] = dict(begin=begin, raw_node_idx=raw, node_idx=g.node_idx, r_begin=g.r_begin,
                                        r_node_idx=g.r_node_idx, rank=rank, dist=dist, sssp_len=sssp_len, sssp_dist=sssp_dist,
                                        age=props["age"], member=props["member"], teen_cnt=props["teen_cnt"], **bc_arrays(bcs))
        if cn is not None:
            fixtures["hand_" + name].update(cn_src=cn["cn_src"], cn_dst=cn["cn_dst"], cn_counts=cn["cn_counts"])
        manifest["hand"][name] = {"N": N, "M": len(edges), "root": root, "pr_iters": it, "tc": T, "tc_cn": cn["tc_cn"] if cn else 0,
                                  "teen_avg_K5_K25_K100": teen_avgs, "conduct_0_4": conducts}
        print("hand %s: pinned (iters=%d T=%d)" % (name, it, T))

    # ---- 3b. the uniform generators (graph_gen.cc:12-105): outputs of the compiled reference, for the host
    #          library's create_uniform_random_graph_new (tests/test_host_cpp.py compares)
    manifest["uniform"] = {}
    for nm, (N, M, seed, xs) in {"uniform_rand_1000_8000": (1000, 8000, 1997, 0), "uniform_rand_77_2000": (77, 2000, 5, 0),
                                 "uniform_xorshift_1000_8000": (1000, 8000, 1997, 1)}.items():
        if xs and R.ref_rand32_min(seed, 2 * M) < 0:
            print("%s: skipped -- the reference's xorshift stream goes negative (r %% N would index out of bounds)" % nm)
            manifest["uniform"][nm] = {"N": N, "M": M, "seed": seed, "xorshift": xs, "skipped": "negative draws: undefined behaviour in the reference"}
            continue
        ub, un = np.zeros(N + 1, np.int32), np.zeros(M, np.int32)
        assert R.ref_uniform_graph(N, M, seed, xs, ub, un) == 0
        fixtures[nm] = dict(begin=ub, node_idx=un)
        manifest["uniform"][nm] = {"N": N, "M": M, "seed": seed, "xorshift": xs, "sha_begin": sha(ub), "sha_node_idx": sha(un)}
        print("%s: pinned" % nm)

    # ---- 4. binary format ----
    begin, raw, snode, rb, rn = ref_graph(R, 256, 4096, 1997, 0.57, 0.19, 0.19, False)
    with tempfile.TemporaryDirectory() as td:
        p_ref = os.path.join(td, "ref.bin")
        p_or = os.path.join(td, "or.bin")
        assert R.ref_store_binary(p_ref.encode(), 256, 4096, begin, snode) == 0
        go = po.Graph(256, begin, snode)
        po.store_binary(p_or, go)
        b_ref = open(p_ref, "rb").read()
        assert b_ref == open(p_or, "rb").read(), "store_binary bytes differ"
        gl = po.load_binary(p_ref)
        assert np.array_equal(gl.begin, begin) and np.array_equal(gl.node_idx, snode)
        assert np.array_equal(gl.r_begin, rb) and np.array_equal(gl.r_node_idx, rn)
        # reference load of the oracle-written file
        N_ = C.c_int32(0)
        M_ = C.c_int32(0)
        lb, ln, lrb, lrn = np.empty(257, np.int32), e32(4096), np.empty(257, np.int32), e32(4096)
        assert R.ref_load_binary(p_or.encode(), C.byref(N_), C.byref(M_), lb.ctypes.data, ln.ctypes.data,
                                 lrb.ctypes.data, lrn.ctypes.data) == 0
        assert np.array_equal(lb, begin) and np.array_equal(ln, snode) and np.array_equal(lrn, rn)
        manifest["bin"] = {"file": "rmat8_ref_store_binary.bin", "N": 256, "M": 4096, "sha256": hashlib.sha256(b_ref).hexdigest()}
        if not args.check:
            open(os.path.join(GOLD, "rmat8_ref_store_binary.bin"), "wb").write(b_ref)
    print("binary format: pinned")

    if not args.check:
        flat = {}
        for k, d in fixtures.items():
            for kk, v in d.items():
                flat[k + "/" + kk] = v
        np.savez_compressed(os.path.join(GOLD, "golden.npz"), **flat)
        json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)
        print("wrote", GOLD)
    print("ORACLE PINNED OK")


if __name__ == "__main__":
    main()
