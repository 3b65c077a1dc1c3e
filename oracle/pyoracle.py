"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.gmo_rmat_edge_list.argtypes = [C.c_int32, C.c_int32, C.c_long, C.c_double, C.c_double, C.c_double,
                                         C.c_int, i32p, i32p, C.POINTER(C.c_int64)]
        L.gmo_create_rmat_graph.argtypes = [C.c_int32, C.c_int32, C.c_long, C.c_double, C.c_double, C.c_double,
                                            C.c_int, i32p, i32p, C.POINTER(C.c_int64)]
        L.gmo_csr_from_edges.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p]
        L.gmo_csr_from_edges.restype = None
        L.gmo_semi_sort.argtypes = [C.c_int32, i32p, i32p]
        L.gmo_semi_sort.restype = None
        L.gmo_make_reverse_edges.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p]
        L.gmo_make_reverse_edges.restype = None
        L.gmo_get_edge_idx_for_src_dest.argtypes = [i32p, i32p, C.c_int32, C.c_int32]
        L.gmo_get_edge_idx_for_src_dest.restype = C.c_int32
        L.gmo_pagerank.argtypes = [C.c_int32, i32p, i32p, i32p, C.c_double, C.c_double, C.c_int32, f64p,
                                   C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.gmo_pagerank.restype = None
        L.gmo_hop_dist.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p, C.c_int, C.POINTER(C.c_int32)]
        L.gmo_hop_dist.restype = None
        L.gmo_sssp.argtypes = [C.c_int32, i32p, i32p, i32p, C.c_int32, i32p, C.c_int, C.POINTER(C.c_int32)]
        L.gmo_sssp.restype = None
        L.gmo_avg_teen_cnt.argtypes = [C.c_int32, i32p, i32p, i32p, i32p, C.c_int32, C.c_int]
        L.gmo_avg_teen_cnt.restype = C.c_float
        L.gmo_conduct.argtypes = [C.c_int32, i32p, i32p, i32p, C.c_int32, C.c_int]
        L.gmo_conduct.restype = C.c_float
        L.gmo_common_nbrs.argtypes = [i32p, i32p, C.c_int32, C.c_int32, i32p, C.c_int64]
        L.gmo_common_nbrs.restype = C.c_int64
        L.gmo_triangle_counting_cn.argtypes = [C.c_int32, i32p, i32p, C.c_int]
        L.gmo_triangle_counting_cn.restype = C.c_int64
        L.gmo_bc.argtypes = [C.c_int32, i32p, i32p, i32p, i32p, i32p, C.c_int32, C.c_int, f32p]
        L.gmo_bc.restype = None
        L.gmo_bfs_queue.argtypes = [C.c_int32, i32p, i32p, C.c_int32, i32p]
        L.gmo_bfs_queue.restype = None
        L.gmo_triangle_counting.argtypes = [C.c_int32, i32p, i32p, C.c_int]
        L.gmo_triangle_counting.restype = C.c_int64
        L.gmo_triangle_counting_merge.argtypes = [C.c_int32, i32p, i32p, i32p, i32p, C.c_int]
        L.gmo_triangle_counting_merge.restype = C.c_int64
        L.gmo_symmetrize.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, i32p]
        L.gmo_symmetrize.restype = C.c_int32
        L.gmo_store_binary.argtypes = [C.c_char_p, C.c_int32, C.c_int32, i32p, i32p]
        L.gmo_load_binary.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
        L.gmo_max_threads.restype = C.c_int

        class R48(C.Structure):
            _fields_ = [("x", C.c_uint64)]
        L.R48 = R48
        L.gmo_srand48.argtypes = [C.POINTER(R48), C.c_long]
        L.gmo_srand48.restype = None
        L.gmo_drand48.argtypes = [C.POINTER(R48)]
        L.gmo_drand48.restype = C.c_double
        _LIB = L
    return _LIB


class Graph:
    """Host CSR in the state load_binary leaves it: semi-sorted + reverse edges."""

    def __init__(self, N, begin, node_idx, r_begin=None, r_node_idx=None):
        self.N = int(N)
        self.M = int(len(node_idx))
        self.begin = np.ascontiguousarray(begin, np.int32)
        self.node_idx = np.ascontiguousarray(node_idx, np.int32)
        self.r_begin = r_begin
        self.r_node_idx = r_node_idx

    def prepare(self):
        L = lib()
        L.gmo_semi_sort(self.N, self.begin, self.node_idx)
        self.r_begin = np.empty(self.N + 1, np.int32)
        self.r_node_idx = np.empty(max(self.M, 1), np.int32)[: self.M]
        self.r_node_idx = np.ascontiguousarray(self.r_node_idx)
        L.gmo_make_reverse_edges(self.N, self.M, self.begin, self.node_idx, self.r_begin, self.r_node_idx)
        return self


def drand48_stream(seed, n):
    L = lib()
    s = L.R48()
    L.gmo_srand48(C.byref(s), seed)
    return np.array([L.gmo_drand48(C.byref(s)) for _ in range(n)], np.float64)


def rmat_edge_list(N, M, seed=1997, a=0.57, b=0.19, c=0.19, permute=False):
    L = lib()
    src = np.empty(max(M, 1), np.int32)[:M].copy()
    dst = np.empty(max(M, 1), np.int32)[:M].copy()
    att = C.c_int64(0)
    rc = L.gmo_rmat_edge_list(N, M, seed, a, b, c, int(permute), src, dst, C.byref(att))
    assert rc == 0, rc
    return src, dst, att.value


def rmat_raw_csr(N, M, seed=1997, a=0.57, b=0.19, c=0.19, permute=False):
    L = lib()
    begin = np.empty(N + 1, np.int32)
    node_idx = np.empty(max(M, 1), np.int32)[:M].copy()
    att = C.c_int64(0)
    rc = L.gmo_create_rmat_graph(N, M, seed, a, b, c, int(permute), begin, node_idx, C.byref(att))
    assert rc == 0, rc
    return begin, node_idx, att.value


def rmat_graph(scale, edge_factor=16, seed=1997, a=0.57, b=0.19, c=0.19, permute=False):
    N = 1 << scale
    M = edge_factor * N
    begin, node_idx, _ = rmat_raw_csr(N, M, seed, a, b, c, permute)
    return Graph(N, begin, node_idx).prepare()


def csr_from_edges(N, src, dst):
    L = lib()
    M = len(src)
    begin = np.empty(N + 1, np.int32)
    node_idx = np.empty(max(M, 1), np.int32)[:M].copy()
    L.gmo_csr_from_edges(N, M, np.ascontiguousarray(src, np.int32), np.ascontiguousarray(dst, np.int32),
                         begin, node_idx)
    return begin, node_idx


def graph_from_edges(N, src, dst):
    begin, node_idx = csr_from_edges(N, src, dst)
    return Graph(N, begin, node_idx).prepare()


def pagerank(g, e=0.001, d=0.85, max_iter=100, nthreads=0):
    L = lib()
    rank = np.empty(max(g.N, 1), np.float64)[: g.N].copy()
    it = C.c_int32(0)
    diff = C.c_double(0)
    L.gmo_pagerank(g.N, g.begin, g.r_begin, g.r_node_idx, e, d, max_iter, rank, nthreads,
                   C.byref(it), C.byref(diff))
    return rank, it.value, diff.value


def hop_dist(g, root=0, nthreads=0):
    L = lib()
    dist = np.empty(max(g.N, 1), np.int32)[: g.N].copy()
    lv = C.c_int32(0)
    L.gmo_hop_dist(g.N, g.begin, g.node_idx, root, dist, nthreads, C.byref(lv))
    return dist, lv.value


def sssp(g, length, root=0, nthreads=0):
    """sssp(G, dist, len, root): len[E] int32 indexed by forward edge slot.  Returns (dist, rounds)."""
    L = lib()
    length = np.ascontiguousarray(length, np.int32)
    assert len(length) == g.M
    dist = np.empty(max(g.N, 1), np.int32)[: g.N].copy()
    rd = C.c_int32(0)
    L.gmo_sssp(g.N, g.begin, g.node_idx, length, root, dist, nthreads, C.byref(rd))
    return dist, rd.value


def avg_teen_cnt(g, age, K, nthreads=0):
    """avg_teen_cnt(G, age, teen_cnt, K) -> (avg as float32, teen_cnt[int32])."""
    age = np.ascontiguousarray(age, np.int32)
    cnt = np.empty(max(g.N, 1), np.int32)[: g.N].copy()
    avg = lib().gmo_avg_teen_cnt(g.N, g.r_begin, g.r_node_idx, age, cnt, int(K), nthreads)
    return np.float32(avg), cnt


def conduct(g, member, num, nthreads=0):
    """conduct(G, member, num) -> float32."""
    member = np.ascontiguousarray(member, np.int32)
    return np.float32(lib().gmo_conduct(g.N, g.begin, g.node_idx, member, int(num), nthreads))


def common_nbrs(g, s, d):
    """Items of gm_common_neighbor_iter(G, s, d), in order."""
    cap = int(g.begin[s + 1] - g.begin[s])
    out = np.zeros(max(cap, 1), np.int32)
    n = lib().gmo_common_nbrs(g.begin, g.node_idx, int(s), int(d), out, cap)
    return out[:n]


def triangle_counting_cn(g, nthreads=0):
    return int(lib().gmo_triangle_counting_cn(g.N, g.begin, g.node_idx, nthreads))


def bc(g, seeds, skip_root=False):
    """comp_BC of apps/src/bc.gm over the seed sequence -> float32 BC[N] (skip_root: upstream's (v != s) filters)."""
    seeds = np.ascontiguousarray(seeds, np.int32)
    out = np.zeros(max(g.N, 1), np.float32)
    lib().gmo_bc(g.N, g.begin, g.node_idx, g.r_begin, g.r_node_idx, seeds, len(seeds), int(bool(skip_root)), out)
    return out[:g.N]


def bfs_queue(g, root=0):
    L = lib()
    dist = np.empty(max(g.N, 1), np.int32)[: g.N].copy()
    L.gmo_bfs_queue(g.N, g.begin, g.node_idx, root, dist)
    return dist


def triangle_counting(g, nthreads=0):
    return int(lib().gmo_triangle_counting(g.N, g.begin, g.node_idx, nthreads))


def triangle_counting_merge(g, nthreads=0):
    return int(lib().gmo_triangle_counting_merge(g.N, g.begin, g.node_idx, g.r_begin, g.r_node_idx, nthreads))


def symmetrize(g):
    L = lib()
    ob = np.empty(g.N + 1, np.int32)
    on = np.empty(max(2 * g.M, 1), np.int32)
    m = L.gmo_symmetrize(g.N, g.M, g.begin, g.node_idx, ob, on)
    return Graph(g.N, ob, on[:m].copy()).prepare()


def store_binary(path, g):
    rc = lib().gmo_store_binary(path.encode(), g.N, g.M, g.begin, g.node_idx)
    assert rc == 0, rc


def load_binary(path):
    L = lib()
    N = C.c_int32(0)
    M = C.c_int32(0)
    rc = L.gmo_load_binary(path.encode(), C.byref(N), C.byref(M), None, None)
    assert rc == 0, rc
    begin = np.empty(N.value + 1, np.int32)
    node_idx = np.empty(max(M.value, 1), np.int32)[: M.value].copy()
    rc = L.gmo_load_binary(path.encode(), C.byref(N), C.byref(M), begin.ctypes.data, node_idx.ctypes.data)
    assert rc == 0, rc
    return Graph(N.value, begin, node_idx).prepare()
