"""ctypes binding of libgmx.so -- the Python host-side mirror of the C ABI in include/gmx.h.

The product path.  There is NO CPU fallback here: if libgmx.so is missing or no
gfx950 device is visible, every compute entry raises.  (The CPU oracle lives in
oracle/ and is only ever loaded by tests / smoke / bench's cpu_baseline leg.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMX_LIB") or os.path.join(_HERE, "libgmx.so")   # GMX_LIB: the debug build of a test

GMX_GRAPH_SORT_ROWS = 0x1
GMX_GRAPH_NO_REVERSE = 0x2
GMX_PR_RELABEL = 0x1
GMX_PR_HOT_LDS = 0x2
GMX_PR_SLICED = 0x4
GMX_PR_COLD_PB = 0x8
INT_MAX = 2147483647


class GmxError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("reserved", C.c_int32), ("last_diff", C.c_double),
                ("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("edges_examined", C.c_int64), ("vertices_reached", C.c_int64), ("edges_reached", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("hbm_bytes", C.c_int64), ("l2_bytes", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32)]


EXPORTS = [
    "gmx_workspace_bytes", "gmx_workspace_release", "gmx_last_error", "gmx_device_count", "gmx_set_device", "gmx_device_info", "gmx_copy_bandwidth",
    "gmx_graph_upload", "gmx_graph_from_edges", "gmx_graph_create_rmat", "gmx_graph_free", "gmx_graph_symmetrize",
    "gmx_graph_num_nodes", "gmx_graph_num_edges", "gmx_graph_download", "gmx_graph_edge_order",
    "gmx_graph_upload_e64", "gmx_graph_download_e64", "gmx_graph_edge_order_e64", "gmx_graph_reverse_edge_map_e64",
    "gmx_pagerank_f64", "gmx_pagerank_f32", "gmx_hop_dist", "gmx_bfs_levels", "gmx_bc", "gmx_sssp", "gmx_avg_teen_cnt", "gmx_conduct", "gmx_triangle_counting", "gmx_triangle_counting_part", "gmx_triangle_counting_cn", "gmx_common_nbrs", "gmx_common_nbr_counts", "gmx_graph_reverse_edge_map",
    "gmx_bfs_create", "gmx_bfs_free", "gmx_bfs_start", "gmx_bfs_step_begin", "gmx_bfs_found_bitmap", "gmx_bfs_step_end",
    "gmx_bfs_download",
    "gmx_pr_create", "gmx_pr_free", "gmx_pr_reset", "gmx_pr_step", "gmx_pr_contrib_slice",
    "gmx_pr_contrib_full", "gmx_pr_exchange_count", "gmx_pr_diff_ptr", "gmx_pr_diff", "gmx_pr_download", "gmx_pr_work",
    "gmx_pr_set_chunks", "gmx_pr_num_chunks", "gmx_pr_chunk_range", "gmx_pr_step_chunk", "gmx_pr_contrib_next_full",
    "gmx_ipc_export", "gmx_ipc_open", "gmx_ipc_close", "gmx_pr_contrib_buffers", "gmx_pr_set_peers",
    "gmx_pr_push_chunk", "gmx_pr_push_current", "gmx_pr_push_join",
    "gmx_pr_packed_info", "gmx_pr_recv_buffers", "gmx_pr_set_peers_packed", "gmx_pr_push_packed", "gmx_pr_unpack", "gmx_pr_recv_list",
    "gmx_pr_gather_classes", "gmx_pr_step_gather", "gmx_pr_push_join_chunk", "gmx_pr_gather_items",
    "gmx_pr_timing", "gmx_pr_kernel_time", "gmx_pr_kernel_name", "gmx_pr_default_options", "gmx_pr_cold_info",
]

_LIB = None
IPC_HANDLE_BYTES = 64   # GMX_IPC_HANDLE_BYTES


def lib():
    """Load libgmx.so (raises if the HIP extension has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GmxError("libgmx.so not built: run `make -C green-marl_amd lib` (or __graft_entry__.build())")
        # PyTorch-ROCm bundles its own HIP runtime (same soname as /opt/rocm's).  Whichever is loaded first
        # serves the whole process; torch must be that one, or its own device discovery fails later
        # ("No HIP GPUs are available") when the multi-GPU driver wraps these buffers in tensors.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.gmx_last_error.restype = C.c_char_p
        L.gmx_workspace_bytes.restype = C.c_int64
        L.gmx_device_count.argtypes = [C.POINTER(C.c_int)]
        L.gmx_set_device.argtypes = [C.c_int]
        L.gmx_device_info.argtypes = [C.POINTER(DeviceInfo)]
        L.gmx_copy_bandwidth.argtypes = [i64, C.c_int, C.POINTER(C.c_double)]
        L.gmx_graph_upload.argtypes = [vp, vp, vp, vp, i64, i64, C.c_uint32, C.POINTER(vp)]
        L.gmx_graph_from_edges.argtypes = [vp, vp, i64, i64, C.c_uint32, C.POINTER(vp)]
        L.gmx_graph_create_rmat.argtypes = [i64, i64, C.c_long, C.c_double, C.c_double, C.c_double, C.c_int,
                                            C.c_uint32, C.POINTER(vp)]
        L.gmx_graph_free.argtypes = [vp]
        L.gmx_graph_symmetrize.argtypes = [vp, C.POINTER(vp)]
        L.gmx_graph_num_nodes.argtypes = [vp]
        L.gmx_graph_num_nodes.restype = i64
        L.gmx_graph_num_edges.argtypes = [vp]
        L.gmx_graph_num_edges.restype = i64
        L.gmx_graph_download.argtypes = [vp, vp, vp, vp, vp]
        L.gmx_graph_edge_order.argtypes = [vp, vp, C.POINTER(C.c_int)]
        L.gmx_pagerank_f64.argtypes = [vp, C.c_double, C.c_double, i32, vp, C.POINTER(Stats)]
        L.gmx_pagerank_f32.argtypes = [vp, C.c_float, C.c_float, i32, vp, C.POINTER(Stats)]
        L.gmx_hop_dist.argtypes = [vp, i32, vp, C.POINTER(Stats)]
        L.gmx_common_nbrs.argtypes = [vp, i32, i32, vp, i64, C.POINTER(i64)]
        L.gmx_common_nbr_counts.argtypes = [vp, vp, vp, i64, vp]
        L.gmx_triangle_counting_cn.argtypes = [vp, C.POINTER(i64), C.POINTER(Stats)]
        L.gmx_bfs_levels.argtypes = [vp, i32, vp, C.POINTER(i32)]
        L.gmx_bc.argtypes = [vp, vp, i32, C.c_int, vp, C.POINTER(Stats)]
        L.gmx_triangle_counting.argtypes = [vp, C.POINTER(i64), C.POINTER(Stats)]
        L.gmx_sssp.argtypes = [vp, i32, vp, vp, C.POINTER(Stats)]
        L.gmx_avg_teen_cnt.argtypes = [vp, vp, i32, vp, C.POINTER(C.c_float), C.POINTER(Stats)]
        L.gmx_conduct.argtypes = [vp, vp, i32, C.POINTER(C.c_float), C.POINTER(Stats)]
        L.gmx_graph_reverse_edge_map.argtypes = [vp, vp]
        L.gmx_graph_upload_e64.argtypes = [vp, vp, vp, vp, i64, i64, C.c_uint32, C.POINTER(vp)]
        L.gmx_graph_download_e64.argtypes = [vp, vp, vp, vp, vp]
        L.gmx_graph_edge_order_e64.argtypes = [vp, vp, C.POINTER(C.c_int)]
        L.gmx_graph_reverse_edge_map_e64.argtypes = [vp, vp]
        L.gmx_triangle_counting_part.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(Stats)]
        L.gmx_bfs_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.gmx_bfs_free.argtypes = [vp]
        L.gmx_bfs_start.argtypes = [vp, i32]
        L.gmx_bfs_step_begin.argtypes = [vp, C.POINTER(C.c_int)]
        L.gmx_bfs_found_bitmap.argtypes = [vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.gmx_bfs_step_end.argtypes = [vp, C.POINTER(i64)]
        L.gmx_bfs_download.argtypes = [vp, vp, C.POINTER(Stats)]
        L.gmx_pr_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(vp)]
        L.gmx_pr_free.argtypes = [vp]
        L.gmx_pr_reset.argtypes = [vp, C.c_double]
        L.gmx_pr_step.argtypes = [vp, vp]
        L.gmx_pr_contrib_slice.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_contrib_full.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_diff_ptr.argtypes = [vp, C.POINTER(vp)]
        L.gmx_ipc_export.argtypes = [vp, vp]
        L.gmx_ipc_open.argtypes = [vp, C.POINTER(vp)]
        L.gmx_ipc_close.argtypes = [vp]
        L.gmx_pr_contrib_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_set_peers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.gmx_pr_push_chunk.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_push_current.argtypes = [vp, vp]
        L.gmx_pr_push_join.argtypes = [vp, vp]
        L.gmx_pr_gather_classes.argtypes = [vp, C.POINTER(C.c_int)]
        L.gmx_pr_step_gather.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_gather_items.argtypes = [vp, C.c_int, C.POINTER(i64)]
        L.gmx_pr_push_join_chunk.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_packed_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.gmx_pr_recv_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_set_peers_packed.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
        L.gmx_pr_push_packed.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_unpack.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_recv_list.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_set_chunks.argtypes = [vp, C.c_int]
        L.gmx_pr_num_chunks.argtypes = [vp, C.POINTER(C.c_int)]
        L.gmx_pr_chunk_range.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i64)]
        L.gmx_pr_step_chunk.argtypes = [vp, C.c_int, vp]
        L.gmx_pr_contrib_next_full.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.gmx_pr_exchange_count.argtypes = [vp, C.POINTER(i64)]
        L.gmx_pr_diff.argtypes = [vp, vp, C.POINTER(C.c_double)]
        L.gmx_pr_download.argtypes = [vp, vp]
        L.gmx_pr_work.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.gmx_pr_cold_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.gmx_pr_default_options.argtypes = [i64, C.c_int]
        L.gmx_pr_default_options.restype = C.c_uint32
        L.gmx_pr_timing.argtypes = [vp, C.c_int]
        L.gmx_pr_kernel_time.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double)]
        L.gmx_pr_kernel_name.argtypes = [vp]
        L.gmx_pr_kernel_name.restype = C.c_char_p
        _LIB = L
    return _LIB


def _ck(status):
    if status != 0:
        raise GmxError("gmx status %d: %s" % (status, lib().gmx_last_error().decode(errors="replace")))


def device_count():
    n = C.c_int(0)
    st = lib().gmx_device_count(C.byref(n))
    return n.value if st == 0 else 0


def require_device():
    if device_count() < 1:
        raise GmxError("no HIP device visible: the gmx hot path has no CPU fallback")


def set_device(i):
    _ck(lib().gmx_set_device(i))


def device_info():
    d = DeviceInfo()
    _ck(lib().gmx_device_info(C.byref(d)))
    return {"name": d.name.decode(), "arch": d.arch.decode(), "compute_units": d.compute_units,
            "clock_mhz": d.clock_mhz, "hbm_bytes": d.hbm_bytes, "l2_bytes": d.l2_bytes,
            "lds_bytes_per_cu": d.lds_bytes_per_cu}


def copy_bandwidth(nbytes=1 << 30, iters=10):
    """Measured device copy rate in GB/s (bytes read + written)."""
    g = C.c_double(0)
    _ck(lib().gmx_copy_bandwidth(nbytes, iters, C.byref(g)))
    return g.value


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


class Graph:
    """Device-resident CSR (+ reverse CSR) in gm_graph's layout."""

    def __init__(self, handle):
        self._h = handle

    # -- constructors ------------------------------------------------------
    @classmethod
    def upload(cls, begin, node_idx, r_begin=None, r_node_idx=None, flags=0):
        require_device()
        begin, node_idx = _i32(begin), _i32(node_idx)
        V, E = len(begin) - 1, len(node_idx)
        h = C.c_void_p()
        rb = _i32(r_begin) if r_begin is not None else None
        rn = _i32(r_node_idx) if r_node_idx is not None else None
        _ck(lib().gmx_graph_upload(begin.ctypes.data, node_idx.ctypes.data if E else None,
                                   rb.ctypes.data if rb is not None else None,
                                   rn.ctypes.data if (rn is not None and E) else None,
                                   V, E, flags, C.byref(h)))
        return cls(h)

    @classmethod
    def from_edges(cls, V, src, dst, flags=0):
        require_device()
        src, dst = _i32(src), _i32(dst)
        h = C.c_void_p()
        E = len(src)
        _ck(lib().gmx_graph_from_edges(src.ctypes.data if E else None, dst.ctypes.data if E else None,
                                       V, E, flags, C.byref(h)))
        return cls(h)

    @classmethod
    def rmat(cls, N, M, seed=1997, a=0.57, b=0.19, c=0.19, permute=False, flags=0):
        require_device()
        h = C.c_void_p()
        _ck(lib().gmx_graph_create_rmat(N, M, seed, a, b, c, int(permute), flags, C.byref(h)))
        return cls(h)

    def symmetrize(self):
        """Undirected simple version (both orientations, duplicates and self loops dropped)."""
        h = C.c_void_p()
        _ck(lib().gmx_graph_symmetrize(self._h, C.byref(h)))
        return Graph(h)

    # -- accessors ---------------------------------------------------------
    @property
    def V(self):
        return lib().gmx_graph_num_nodes(self._h)

    @property
    def E(self):
        return lib().gmx_graph_num_edges(self._h)

    def download(self, reverse=True):
        V, E = self.V, self.E
        begin = np.empty(V + 1, np.int32)
        node_idx = np.empty(E, np.int32)
        rb = rn = None
        if reverse:
            rb = np.empty(V + 1, np.int32)
            rn = np.empty(E, np.int32)
        _ck(lib().gmx_graph_download(self._h, begin.ctypes.data, node_idx.ctypes.data,
                                     rb.ctypes.data if reverse else None, rn.ctypes.data if reverse else None))
        return begin, node_idx, rb, rn

    def free(self):
        if self._h:
            lib().gmx_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- the three kernels (mirror of the generated entry points) ----------
    def pagerank(self, e=0.001, d=0.85, max_iter=100, dtype=np.float64):
        """pagerank(G, e, d, max, G_pg_rank) -- returns (rank, stats)."""
        rank = np.empty(self.V, dtype)
        st = Stats()
        if dtype == np.float64:
            _ck(lib().gmx_pagerank_f64(self._h, e, d, max_iter, rank.ctypes.data, C.byref(st)))
        elif dtype == np.float32:
            _ck(lib().gmx_pagerank_f32(self._h, e, d, max_iter, rank.ctypes.data, C.byref(st)))
        else:
            raise GmxError("dtype must be float32 or float64")
        return rank, st.as_dict()

    def hop_dist(self, root=0):
        """hop_dist(G, G_dist, root) -- returns (dist, stats)."""
        dist = np.empty(self.V, np.int32)
        st = Stats()
        _ck(lib().gmx_hop_dist(self._h, root, dist.ctypes.data, C.byref(st)))
        return dist, st.as_dict()

    def bfs_levels(self, root=0):
        """prepare(root) + do_bfs_forward() of the BFS object -> (level[int16], unvisited = -2; number of levels)."""
        lv = np.zeros(self.V, np.int16)
        n = C.c_int32(0)
        _ck(lib().gmx_bfs_levels(self._h, int(root), lv.ctypes.data, C.byref(n)))
        return lv, n.value

    def bc(self, seeds, skip_root=False):
        """comp_BC(G, BC, Seeds) of bc.gm -> (BC[float32], stats)."""
        seeds = _i32(seeds)
        out = np.zeros(self.V, np.float32)
        st = Stats()
        _ck(lib().gmx_bc(self._h, seeds.ctypes.data if len(seeds) else None, len(seeds), int(bool(skip_root)), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def sssp(self, length, root=0):
        """sssp(G, dist, len, root): length[E] int32 by forward edge slot -- returns (dist[int32], stats)."""
        length = _i32(length)
        assert len(length) == self.E
        dist = np.zeros(self.V, np.int32)
        st = Stats()
        _ck(lib().gmx_sssp(self._h, int(root), length.ctypes.data if self.E else None, dist.ctypes.data, C.byref(st)))
        return dist, st.as_dict()

    def avg_teen_cnt(self, age, K):
        """avg_teen_cnt(G, age, teen_cnt, K) -- returns (avg float32, teen_cnt[int32], stats)."""
        age = _i32(age)
        assert len(age) == self.V
        cnt = np.zeros(self.V, np.int32)
        avg, st = C.c_float(0), Stats()
        _ck(lib().gmx_avg_teen_cnt(self._h, age.ctypes.data if self.V else None, int(K), cnt.ctypes.data if self.V else None,
                                   C.byref(avg), C.byref(st)))
        return np.float32(avg.value), cnt, st.as_dict()

    def conduct(self, member, num):
        """conduct(G, member, num) -- returns (float32, stats)."""
        member = _i32(member)
        assert len(member) == self.V
        res, st = C.c_float(0), Stats()
        _ck(lib().gmx_conduct(self._h, member.ctypes.data if self.V else None, int(num), C.byref(res), C.byref(st)))
        return np.float32(res.value), st.as_dict()

    def edge_order(self):
        """e_idx2idx after an upload with GMX_GRAPH_SORT_ROWS: uploaded slot of every slot of the sorted rows;
        None when the upload was already in order (identity)."""
        out = np.zeros(self.E, np.int32)
        ident = C.c_int(1)
        _ck(lib().gmx_graph_edge_order(self._h, out.ctypes.data if self.E else None, C.byref(ident)))
        return None if ident.value else out

    def reverse_edge_map(self):
        """e_rev2idx: forward slot mirrored by each reverse-CSR slot."""
        out = np.zeros(self.E, np.int32)
        _ck(lib().gmx_graph_reverse_edge_map(self._h, out.ctypes.data))
        return out

    def common_nbrs(self, s, d):
        """The items gm_common_neighbor_iter(G, s, d) yields, in order."""
        n = C.c_int64(0)
        _ck(lib().gmx_common_nbrs(self._h, int(s), int(d), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.int32)
        _ck(lib().gmx_common_nbrs(self._h, int(s), int(d), out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def common_nbr_counts(self, src, dst):
        src, dst = _i32(src), _i32(dst)
        out = np.zeros(max(len(src), 1), np.int64)
        _ck(lib().gmx_common_nbr_counts(self._h, src.ctypes.data, dst.ctypes.data, len(src), out.ctypes.data))
        return out[:len(src)]

    def triangle_counting_cn(self):
        """Triangle counting written with the common-neighbour iterator -- returns (T, stats)."""
        t, st = C.c_int64(0), Stats()
        _ck(lib().gmx_triangle_counting_cn(self._h, C.byref(t), C.byref(st)))
        return t.value, st.as_dict()

    def triangle_counting(self, part=0, nparts=1):
        """triangle_counting(G) -- returns (T, stats); with nparts > 1 the share of one part of the edge slots."""
        t = C.c_int64(0)
        st = Stats()
        _ck(lib().gmx_triangle_counting_part(self._h, part, nparts, C.byref(t), C.byref(st)))
        return t.value, st.as_dict()


def default_pr_options(V, nranks=1):
    return int(lib().gmx_pr_default_options(V, nranks))


class DevArray:
    """A raw device pointer exposed through __cuda_array_interface__ so that torch can wrap it
    (torch.as_tensor(DevArray(...), device='cuda')) for torch.distributed collectives."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class BfsState:
    """Device-resident hop_dist stepping state (gmx_bfs_*) for one rank of nranks (replicated graph)."""

    def __init__(self, graph, rank=0, nranks=1):
        self.graph = graph
        self.rank, self.nranks = rank, nranks
        h = C.c_void_p()
        _ck(lib().gmx_bfs_create(graph._h, rank, nranks, C.byref(h)))
        self._h = h

    def start(self, root):
        _ck(lib().gmx_bfs_start(self._h, int(root)))

    def step_begin(self):
        """Runs the local part of the next level; True if the found-bitmap slices have to be exchanged."""
        need = C.c_int(0)
        _ck(lib().gmx_bfs_step_begin(self._h, C.byref(need)))
        return bool(need.value)

    def found_bitmap(self):
        """(whole bitmap as DevArray of int64 words, slice offset, slice length) -- words of 64 vertices."""
        p, tot, off, n = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _ck(lib().gmx_bfs_found_bitmap(self._h, C.byref(p), C.byref(tot), C.byref(off), C.byref(n)))
        return DevArray(p.value, tot.value, "<i8"), off.value, n.value

    def step_end(self):
        n = C.c_int64(0)
        _ck(lib().gmx_bfs_step_end(self._h, C.byref(n)))
        return n.value

    def download(self):
        out = np.zeros(self.graph.V, np.int32)
        st = Stats()
        _ck(lib().gmx_bfs_download(self._h, out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def free(self):
        if self._h:
            lib().gmx_bfs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Handles(list):
    """ipc_handles() of a rank: the replica handles as list items, the packed-exchange extras as attributes (picklable)."""

    def __init__(self, *a):
        super().__init__(*a)
        self.landing = []
        self.packed = None

    def __reduce__(self):
        return (_rebuild_handles, (list(self), self.landing, self.packed))


def _rebuild_handles(items, landing, packed):
    h = _Handles(items)
    h.landing, h.packed = landing, packed
    return h


class PageRankState:
    """Device-resident PageRank stepping state (gmx_pr_*) for one rank of nranks."""

    def __init__(self, graph, elem_bytes=4, rank=0, nranks=1, options=GMX_PR_RELABEL):
        self.graph = graph
        self.elem = elem_bytes
        self.rank, self.nranks = rank, nranks
        h = C.c_void_p()
        _ck(lib().gmx_pr_create(graph._h, elem_bytes, rank, nranks, options, C.byref(h)))
        self._h = h

    def reset(self, d=0.85):
        _ck(lib().gmx_pr_reset(self._h, d))

    def step(self, stream=None):
        _ck(lib().gmx_pr_step(self._h, stream))

    # ---- peer push (gmx.h: exchange by direct copies into the peers' replicas, packed when the plan has the lists) ----
    def packed_info(self):
        """{"send": [...], "recv": [...], "offset": [...]} (elements per peer) of the packed exchange, or None when the
        plan has no lists (one rank, or not the degree order)."""
        n = self.nranks
        a, b, c = (C.c_int64 * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
        if lib().gmx_pr_packed_info(self._h, a, b, c) != 0:
            return None
        return {"send": list(a), "recv": list(b), "offset": list(c)}

    def _buffers(self, landing):
        b0, b1, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        _ck((lib().gmx_pr_recv_buffers if landing else lib().gmx_pr_contrib_buffers)(self._h, C.byref(b0), C.byref(b1), C.byref(n)))
        return b0, b1

    def ipc_handles(self):
        """What the other ranks need to push into this one: hipIpc handles (bytes) of the two contribution replicas
        [0], [1]; with a packed plan also ["landing"] (handles of the two landing zones) and ["packed"] (packed_info())."""
        out = _Handles()
        for b in self._buffers(False):
            h = C.create_string_buffer(IPC_HANDLE_BYTES)
            _ck(lib().gmx_ipc_export(b, h))
            out.append(h.raw)
        info = self.packed_info()
        if info is not None and os.environ.get("GMX_PUSH_PACKED", "1") != "0":
            out.packed = info
            for b in self._buffers(True):
                h = C.create_string_buffer(IPC_HANDLE_BYTES)
                _ck(lib().gmx_ipc_export(b, h))
                out.landing.append(h.raw)
        return out

    def _register_peers(self, ptrs, land, infos):
        _ck(lib().gmx_pr_set_peers(self._h, ptrs[0], ptrs[1]))
        self.packed_push = False
        if land is not None:
            off, cnt = (C.c_int64 * self.nranks)(), (C.c_int64 * self.nranks)()
            for r in range(self.nranks):
                if r != self.rank:
                    off[r], cnt[r] = infos[r]["offset"][self.rank], infos[r]["recv"][self.rank]
            _ck(lib().gmx_pr_set_peers_packed(self._h, land[0], land[1], off, cnt))
            self.packed_push = True

    def set_peers(self, handles):
        """handles[r] = rank r's ipc_handles() (own entry ignored): map them and register the pointers.  The packed
        push is used when every rank offers landing zones."""
        ptrs = [(C.c_void_p * self.nranks)(), (C.c_void_p * self.nranks)()]
        packed = self.packed_info() is not None and all(getattr(h, "packed", None) is not None for h in handles)
        land = [(C.c_void_p * self.nranks)(), (C.c_void_p * self.nranks)()] if packed else None
        self._peer_maps = []
        for r, hs in enumerate(handles):
            if r == self.rank:
                continue
            for b in (0, 1):
                for dst, src in ((ptrs, hs), (land, hs.landing if packed else None)):
                    if dst is None:
                        continue
                    q = C.c_void_p()
                    _ck(lib().gmx_ipc_open(C.create_string_buffer(src[b], IPC_HANDLE_BYTES), C.byref(q)))
                    dst[b][r] = q.value
                    self._peer_maps.append(q.value)
        self._register_peers(ptrs, land, [getattr(h, "packed", None) for h in handles])

    def set_peers_local(self, states):
        """The same wiring for rank states that live in THIS process (tests, one-GPU rehearsals): raw device pointers."""
        ptrs = [(C.c_void_p * self.nranks)(), (C.c_void_p * self.nranks)()]
        infos = [s.packed_info() for s in states]
        packed = all(i is not None for i in infos) and os.environ.get("GMX_PUSH_PACKED", "1") != "0"
        land = [(C.c_void_p * self.nranks)(), (C.c_void_p * self.nranks)()] if packed else None
        for r, st in enumerate(states):
            if r == self.rank:
                continue
            for dst, bufs in ((ptrs, st._buffers(False)), (land, st._buffers(True) if packed else None)):
                if dst is not None:
                    dst[0][r], dst[1][r] = bufs[0].value, bufs[1].value
        self._register_peers(ptrs, land, infos)

    def push_chunk(self, chunk, stream=None):
        if getattr(self, "packed_push", False):
            _ck(lib().gmx_pr_push_packed(self._h, int(chunk), stream))
        else:
            _ck(lib().gmx_pr_push_chunk(self._h, chunk, stream))

    def push_current(self, stream=None):
        if getattr(self, "packed_push", False):
            _ck(lib().gmx_pr_push_packed(self._h, -1, stream))
        else:
            _ck(lib().gmx_pr_push_current(self._h, stream))

    def unpack(self, chunk=-1, stream=None):
        """After the barrier that follows the pushes: scatter what the peers packed (chunk, or -1 = everything) into the
        replica the next step reads.  Nothing to do for the plain push."""
        if getattr(self, "packed_push", False):
            _ck(lib().gmx_pr_unpack(self._h, int(chunk), stream))

    def recv_list(self, r):
        """DevArray of the positions (inside rank r's range) this rank reads (packed plans)."""
        p, n = C.c_void_p(), C.c_int64(0)
        _ck(lib().gmx_pr_recv_list(self._h, int(r), C.byref(p), C.byref(n)))
        return DevArray(p.value, n.value, "<i4")

    def exchange_bytes(self):
        """Bytes this rank sends per step: the packed lists if they are in use, else the exchanged prefix to every peer."""
        if getattr(self, "packed_push", False):
            return sum(self.packed_info()["send"]) * self.elem
        return self.exchange_count() * (self.nranks - 1) * self.elem

    def push_join(self, stream=None):
        _ck(lib().gmx_pr_push_join(self._h, stream))

    def push_join_chunk(self, chunk, stream=None):
        """`stream` waits for the copies of one chunk only."""
        _ck(lib().gmx_pr_push_join_chunk(self._h, int(chunk), stream))

    def gather_classes(self):
        """2 when the step can be pipelined (every in-edge binned: step_gather(0/1) ahead of step_chunk), else 0."""
        n = C.c_int(0)
        _ck(lib().gmx_pr_gather_classes(self._h, C.byref(n)))
        return n.value

    def gather_items(self, tile_class):
        n = C.c_int64(0)
        _ck(lib().gmx_pr_gather_items(self._h, int(tile_class), C.byref(n)))
        return n.value

    def step_gather(self, tile_class, stream=None):
        _ck(lib().gmx_pr_step_gather(self._h, int(tile_class), stream))

    def set_chunks(self, chunks):
        _ck(lib().gmx_pr_set_chunks(self._h, int(chunks)))
        return self.num_chunks()

    def num_chunks(self):
        n = C.c_int(0)
        _ck(lib().gmx_pr_num_chunks(self._h, C.byref(n)))
        return n.value

    def chunk_range(self, chunk):
        off, cnt = C.c_int64(0), C.c_int64(0)
        _ck(lib().gmx_pr_chunk_range(self._h, chunk, C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def step_chunk(self, chunk, stream=None):
        _ck(lib().gmx_pr_step_chunk(self._h, chunk, stream))

    def contrib_next_full(self):
        p, n = C.c_void_p(), C.c_int64(0)
        _ck(lib().gmx_pr_contrib_next_full(self._h, C.byref(p), C.byref(n)))
        return DevArray(p.value, n.value, self._typestr())

    def diff(self, stream=None):
        v = C.c_double(0)
        _ck(lib().gmx_pr_diff(self._h, stream, C.byref(v)))
        return v.value

    def _typestr(self):
        return "<f4" if self.elem == 4 else "<f8"

    def contrib_slice(self):
        p, n = C.c_void_p(), C.c_int64(0)
        _ck(lib().gmx_pr_contrib_slice(self._h, C.byref(p), C.byref(n)))
        return DevArray(p.value, n.value, self._typestr())

    def contrib_full(self):
        p, n = C.c_void_p(), C.c_int64(0)
        _ck(lib().gmx_pr_contrib_full(self._h, C.byref(p), C.byref(n)))
        return DevArray(p.value, n.value, self._typestr())

    def exchange_count(self):
        n = C.c_int64(0)
        _ck(lib().gmx_pr_exchange_count(self._h, C.byref(n)))
        return n.value

    def diff_dev(self):
        p = C.c_void_p()
        _ck(lib().gmx_pr_diff_ptr(self._h, C.byref(p)))
        return DevArray(p.value, 1, "<f8")

    def download(self, out=None):
        V = self.graph.V
        dt = np.float32 if self.elem == 4 else np.float64
        if out is None:
            out = np.zeros(V, dt)
        _ck(lib().gmx_pr_download(self._h, out.ctypes.data))
        return out

    def timing(self, enable=True):
        _ck(lib().gmx_pr_timing(self._h, int(enable)))

    def kernel_time(self):
        """(launches, mean_ms) of the dominant kernel since timing(True), hipEvents on its stream."""
        n, ms = C.c_int32(0), C.c_double(0)
        _ck(lib().gmx_pr_kernel_time(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_name(self):
        return lib().gmx_pr_kernel_name(self._h).decode()

    def cold_info(self):
        """(hot ids per rank range, binned edges, items of phase 2) of the binned part; hot_ids = -1 without one."""
        t, e, p = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _ck(lib().gmx_pr_cold_info(self._h, C.byref(t), C.byref(e), C.byref(p)))
        return {"hot_ids": t.value, "cold_edges": e.value, "padded_items": p.value}

    def work(self):
        e, r, b = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _ck(lib().gmx_pr_work(self._h, C.byref(e), C.byref(r), C.byref(b)))
        return {"edges": e.value, "rows": r.value, "algorithmic_bytes": b.value}

    def free(self):
        if self._h:
            lib().gmx_pr_free(self._h)     # drains the copy streams first
            self._h = None
            for q in getattr(self, "_peer_maps", []):
                lib().gmx_ipc_close(q)
            self._peer_maps = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
