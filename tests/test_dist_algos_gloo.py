"""CPU: the N > 1 orchestration of hop_dist and triangle_counting (green-marl_amd/dist_algos.py) over gloo.
The local work is done by test-owned numpy engines with the stepping interface of gmx.BfsState; what is under
test is the exchange / termination logic the GPU ranks run unchanged."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyoracle as po

INT_MAX = 2147483647


class NumpyBfsEngine:
    """hop_dist with every level run bottom-up over the rank's vertex range (found bitmap in 64-bit words)."""

    def __init__(self, g, rank, world):
        self.g, self.rank, self.world = g, rank, world
        w = (g.N + 63) // 64
        self.slice_words = max((w + world - 1) // world, 1)
        self.words = self.slice_words * world
        self.found = torch.zeros(self.words, dtype=torch.int64)
        self.src_of = np.repeat(np.arange(g.N), np.diff(g.r_begin))   # destination of every reverse slot

    def start(self, root):
        self.dist = np.full(self.g.N, INT_MAX, np.int32)
        self.frontier = np.zeros(self.g.N, bool)
        self.level = 0
        self.count = 0
        if 0 <= root < self.g.N:
            self.dist[root] = 0
            self.frontier[root] = True
            self.count = 1

    def step_begin(self):
        if self.count == 0:
            return False
        lo = self.rank * self.slice_words * 64
        hi = min(lo + self.slice_words * 64, self.g.N)
        hit = np.zeros(self.g.N, bool)
        sel = self.frontier[self.g.r_node_idx] & (self.dist[self.src_of] == INT_MAX)
        hit[self.src_of[sel]] = True
        bits = np.zeros(self.words * 64, bool)
        bits[lo:hi] = hit[lo:hi]
        mine = np.packbits(bits.reshape(-1, 64)[:, ::-1], axis=1).view(">u8").astype(np.uint64).ravel()
        lo_w = self.rank * self.slice_words
        self.found[lo_w:lo_w + self.slice_words] = torch.from_numpy(mine[lo_w:lo_w + self.slice_words].view(np.int64).copy())
        return self.world > 1

    def found_bitmap(self):
        return self.found, self.rank * self.slice_words, self.slice_words

    def step_end(self):
        if self.count == 0:
            return 0
        w = self.found.numpy().view(np.uint64)
        bits = ((w[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).ravel()[:self.g.N]
        self.dist[bits] = self.level + 1
        self.frontier = bits
        self.count = int(bits.sum())
        self.level += 1
        return self.count


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scale, root, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "green-marl_amd"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, os.path.abspath(p))
    from dist_algos import DistHopDist, dist_triangle_counting
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = po.rmat_graph(scale, permute=False)
    eng = NumpyBfsEngine(g, rank, world)
    drv = DistHopDist(eng)
    levels = drv.run(root)
    np.save(os.path.join(out_dir, "dist%d.npy" % rank), eng.dist)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([levels, drv.exchanges]))
    # triangle counting: any split of the slots whose shares add up; here the true count dealt unevenly
    total = 1000003
    share = lambda p, n: total // n + (1 if p < total % n else 0) + (7 if p == 0 else 0) - (7 if p == n - 1 else 0)
    np.save(os.path.join(out_dir, "tc%d.npy" % rank), np.array([dist_triangle_counting(share)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_dist_hop_dist_and_tc_gloo(tmp_path, world):
    scale, root = 11, 0
    mp.spawn(_worker, args=(world, _free_port(), scale, root, str(tmp_path)), nprocs=world, join=True)
    g = po.rmat_graph(scale, permute=False)
    want = po.hop_dist(g, root)[0]
    depth = int(want[want != INT_MAX].max())
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("dist%d.npy" % r)), want)
        levels, exchanges = np.load(tmp_path / ("meta%d.npy" % r))
        assert levels == depth and exchanges == depth + 1     # one all-gather per level, the empty last one included
        assert np.load(tmp_path / ("tc%d.npy" % r))[0] == 1000003


def test_single_process_world1():
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "green-marl_amd"))
    from dist_algos import DistHopDist, dist_triangle_counting
    g = po.rmat_graph(10, permute=False)
    eng = NumpyBfsEngine(g, 0, 1)
    DistHopDist(eng).run(0)
    assert np.array_equal(eng.dist, po.hop_dist(g, 0)[0])
    assert dist_triangle_counting(lambda p, n: 42) == 42
