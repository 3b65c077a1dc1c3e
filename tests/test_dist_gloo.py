"""CPU: the N > 1 orchestration (green-marl_amd/dist_pagerank.py) over gloo, world_size 2 and 3.
The local sweep is a test-owned numpy engine (the oracle's arithmetic on a 1-D vertex slice);
what is under test is the exchange / diff / termination logic the GPU ranks run unchanged."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pyoracle as po


class NumpyEngine:
    """Owns rows [lo, hi) of an equal 1-D split; same stepping interface as dist_pagerank.GmxEngine."""

    def __init__(self, g, rank, world):
        self.g, self.rank, self.world = g, rank, world
        self.slice = (g.N + world - 1) // world
        self.lo = rank * self.slice
        self.hi = min(g.N, self.lo + self.slice)
        self.outdeg = np.diff(g.begin).astype(np.float64)
        self.dst = np.repeat(np.arange(g.N), np.diff(g.r_begin))
        self.sel = (self.dst >= self.lo) & (self.dst < self.hi)
        self.contrib = [torch.zeros(self.slice * world, dtype=torch.float64) for _ in range(2)]
        self.cur = 0
        self.rank_v = np.zeros(max(self.hi - self.lo, 0))
        self._diff = torch.zeros(1, dtype=torch.float64)

    def _contrib_of(self, r):
        od = self.outdeg[self.lo:self.hi]
        return np.where(od > 0, r / np.maximum(od, 1), 0.0)

    def reset(self, d):
        self.d = d
        self.cur = 0
        self.rank_v[:] = 1.0 / self.g.N
        # only the owned range is written (as gmx_pr_reset does): a faster peer may already have pushed its
        # slice into this replica
        self.contrib[0][self.lo:self.hi] = torch.from_numpy(self._contrib_of(self.rank_v))

    def step(self):
        c = self.contrib[self.cur].numpy()
        sums = np.bincount(self.dst[self.sel] - self.lo, weights=c[self.g.r_node_idx[self.sel]],
                           minlength=self.hi - self.lo)
        val = (1 - self.d) / self.g.N + self.d * sums
        self._diff[0] = np.abs(val - self.rank_v).sum()
        self.rank_v = val
        nxt = self.contrib[1 - self.cur]
        nxt[self.lo:self.hi] = torch.from_numpy(self._contrib_of(val))
        self.cur = 1 - self.cur

    # ---- row chunks (gmx_pr_set_chunks / _step_chunk / _chunk_range / _contrib_next_full) ----
    nchunks = 1

    def set_chunks(self, chunks):
        self.nchunks = chunks
        return chunks

    def num_chunks(self):
        return self.nchunks

    def chunk_range(self, c):
        step = (self.slice + self.nchunks - 1) // self.nchunks
        lo, hi = min(c * step, self.slice), min((c + 1) * step, self.slice)
        return lo, hi - lo

    def step_chunk(self, c):
        off, cnt = self.chunk_range(c)
        lo, hi = min(self.lo + off, self.hi), min(self.lo + off + cnt, self.hi)
        src = self.contrib[self.cur].numpy()
        sel = (self.dst >= lo) & (self.dst < hi)
        sums = np.bincount(self.dst[sel] - lo, weights=src[self.g.r_node_idx[sel]], minlength=hi - lo)
        val = (1 - self.d) / self.g.N + self.d * sums
        if c == 0:
            self._diff[0] = 0.0
        self._diff[0] += np.abs(val - self.rank_v[lo - self.lo:hi - self.lo]).sum()
        self.rank_v[lo - self.lo:hi - self.lo] = val
        od = self.outdeg[lo:hi]
        self.contrib[1 - self.cur][lo:hi] = torch.from_numpy(np.where(od > 0, val / np.maximum(od, 1), 0.0))
        if c == self.nchunks - 1:
            self.cur = 1 - self.cur

    def contrib_next_full(self):
        return self.contrib[1 - self.cur]

    def contrib_slice(self):
        return self.contrib[self.cur][self.lo:self.lo + self.slice]

    def contrib_full(self):
        return self.contrib[self.cur]

    def diff_tensor(self):
        return self._diff


class SharedFileEngine(NumpyEngine):
    """NumpyEngine whose replicas live in files every rank can map -- the CPU stand-in for hipIpc-mapped HBM.
    Implements the peer-push interface of dist_pagerank.GmxEngine (ipc_handles / set_peers / push_*)."""

    def __init__(self, g, rank, world, tmp_dir):
        super().__init__(g, rank, world)
        self.paths = [os.path.join(tmp_dir, "replica_r%d_b%d.bin" % (rank, b)) for b in (0, 1)]
        maps = [np.memmap(p, dtype=np.float64, mode="w+", shape=(self.slice * world,)) for p in self.paths]
        self.contrib = [torch.from_numpy(m) for m in maps]
        self.peers = None

    def ipc_handles(self):
        return list(self.paths)

    def set_peers(self, handles):
        self.peers = [None if r == self.rank else
                      [np.memmap(p, dtype=np.float64, mode="r+", shape=(self.slice * self.world,)) for p in hs]
                      for r, hs in enumerate(handles)]

    def _push(self, b, off, cnt):
        lo = self.rank * self.slice + off
        src = self.contrib[b].numpy()[lo:lo + cnt]
        for r, maps in enumerate(self.peers):
            if maps is not None:
                maps[b][lo:lo + cnt] = src
                maps[b].flush()

    def step_chunk(self, c):
        if c == 0:
            self.step_next = 1 - self.cur
        super().step_chunk(c)

    def push_chunk(self, c):
        self._push(self.step_next, *self.chunk_range(c))

    def push_current(self):
        self._push(self.cur, 0, self.slice)

    def push_join(self):
        pass


class PipelinedFileEngine(SharedFileEngine):
    """SharedFileEngine whose step reads the replica only in two gather calls, like the binned GPU step
    (gmx_pr_step_gather): class 0 takes every rank's LAST chunk (the hub piece), class 1 the rest; the chunks are
    then computed from that snapshot.  A gather issued before its pieces have landed gives wrong ranks."""

    def gather_classes(self):
        return 2

    def push_join_chunk(self, c, stream=None):
        pass

    def step_gather(self, cls):
        if cls == 0:
            self.step_next = 1 - self.cur
            self.snap = np.full(self.slice * self.world, np.nan)
        off, cnt = self.chunk_range(self.nchunks - 1)
        hub = np.zeros(self.slice * self.world, dtype=bool)
        for r in range(self.world):
            hub[r * self.slice + off:r * self.slice + off + cnt] = True
        pick = hub if cls == 0 else ~hub
        self.snap[pick] = self.contrib[self.cur].numpy()[pick]
        self.gathered = getattr(self, "gathered", 0) | (1 << cls)

    def step_chunk(self, c):
        if c == 0:
            for cls in (0, 1):      # not driven through the pipelined order: gather here, like gmx_pr_step_chunk(0)
                if not getattr(self, "gathered", 0) & (1 << cls):
                    self.step_gather(cls)
            self.gathered = 0
        live, cur = self.contrib[self.cur], self.cur
        self.contrib[cur] = torch.from_numpy(self.snap)     # what the gathers saw, not what is there now
        try:
            NumpyEngine.step_chunk(self, c)                 # (flips self.cur after the last chunk)
        finally:
            self.contrib[cur] = live


class PackedFileEngine(PipelinedFileEngine):
    """The packed push ("send only what is read", gmx_pr_push_packed / gmx_pr_unpack) on files: per peer the sorted
    positions of its range this rank reads, a landing zone per replica parity that the peers write their packed pieces
    into, and an unpack that scatters them into the replica.  The replicas start as NaN outside the owned range, so a
    position that was needed but never sent -- or a gather that ran before its unpack -- poisons the ranks."""

    def __init__(self, g, rank, world, tmp_dir):
        super().__init__(g, rank, world, tmp_dir)
        for c in self.contrib:
            c[:] = float("nan")
            c[g.N:] = 0.0            # (padding behind the last vertex: never a source)
        sl = self.slice
        # reads[q][r]: positions inside rank r's range that rank q reads (the sources of q's in-edges)
        self.reads = []
        for q in range(world):
            qlo, qhi = q * sl, min(g.N, (q + 1) * sl)
            src = np.unique(g.r_node_idx[(self.dst >= qlo) & (self.dst < qhi)])
            self.reads.append([src[(src >= r * sl) & (src < (r + 1) * sl)] - r * sl for r in range(world)])
        self.roff = np.concatenate([[0], np.cumsum([len(x) for x in self.reads[rank]])]).astype(np.int64)
        self.lpaths = [os.path.join(tmp_dir, "landing_r%d_b%d.bin" % (rank, b)) for b in (0, 1)]
        self.landing = [np.memmap(p, dtype=np.float64, mode="w+", shape=(max(int(self.roff[-1]), 1),)) for p in self.lpaths]
        for m in self.landing:
            m[:] = np.nan

    def packed(self):
        return True

    def recv_list(self, r):
        return torch.from_numpy(self.reads[self.rank][r].astype(np.int64))

    def ipc_handles(self):
        return {"replica": list(self.paths), "landing": list(self.lpaths), "size": int(self.roff[-1])}

    def set_peers(self, handles):
        self.peer_landing = [None if r == self.rank else
                             [np.memmap(p, dtype=np.float64, mode="r+", shape=(max(h["size"], 1),)) for p in h["landing"]]
                             for r, h in enumerate(handles)]
        # where my segment starts in peer q's landing zone: q's recv offsets, which I can compute like q does
        self.my_off = [int(sum(len(x) for x in self.reads[q][:self.rank])) for q in range(self.world)]

    def _push(self, b, off, cnt):
        own = self.contrib[b].numpy()[self.rank * self.slice:(self.rank + 1) * self.slice]
        for q in range(self.world):
            if q == self.rank:
                continue
            lst = self.reads[q][self.rank]
            a, e = np.searchsorted(lst, off), np.searchsorted(lst, off + cnt)
            self.peer_landing[q][b][self.my_off[q] + a:self.my_off[q] + e] = own[lst[a:e]]
            self.peer_landing[q][b].flush()

    def unpack(self, chunk=-1):
        off, cnt = (0, self.slice) if chunk < 0 else self.chunk_range(chunk)
        b = self.cur
        dst = self.contrib[b].numpy()
        for r in range(self.world):
            if r == self.rank:
                continue
            lst = self.reads[self.rank][r]
            a, e = np.searchsorted(lst, off), np.searchsorted(lst, off + cnt)
            dst[r * self.slice + lst[a:e]] = self.landing[b][self.roff[r] + a:self.roff[r] + e]

    def step_gather(self, cls):
        super().step_gather(cls)
        # (the owned range and the listed positions are the only finite entries a step may meet)

    def step_chunk(self, c):
        super().step_chunk(c)
        assert np.isfinite(self.rank_v).all(), "a position this rank reads had not been unpacked when it was gathered"


class BrokenPushEngine(SharedFileEngine):
    """Peer copies that silently land nowhere: DistPageRank's one-time check must notice and fall back."""

    def _push(self, b, off, cnt):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scale, out_dir, chunks=1, push=False, broken=False, pipelined=False, packed=False):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "..", "green-marl_amd"), os.path.join(here, "..", "oracle")):
        sys.path.insert(0, os.path.abspath(p))
    from dist_pagerank import DistPageRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = po.rmat_graph(scale, permute=True)
    cls = BrokenPushEngine if broken else PackedFileEngine if packed else PipelinedFileEngine if pipelined else SharedFileEngine
    eng = cls(g, rank, world, out_dir) if push else NumpyEngine(g, rank, world)
    eng.set_chunks(chunks)
    pr = DistPageRank(eng, exchange="push", barrier="host") if push else DistPageRank(eng)
    assert pr.exchange == ("push" if push else "collective")
    if pipelined:
        calls = []
        orig = pr._step_pipelined
        pr._step_pipelined = lambda: (calls.append(1), orig())[1]
    cnt, diff = pr.run(0.001, 0.85, 100)
    if pipelined:
        assert (len(calls) == cnt) == (chunks == 2)      # the pipelined order is what ran (two chunks), else the plain one
    assert pr.exchange == ("push" if push and not broken else "collective")   # broken copies: fell back
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), eng.rank_v)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([cnt, diff, eng.lo, eng.hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks,push,broken,pipelined,packed", [
    (2, 1, False, False, False, False), (3, 1, False, False, False, False), (2, 4, False, False, False, False), (3, 3, False, False, False, False),
    (2, 1, True, False, False, False), (3, 2, True, False, False, False), (2, 2, True, True, False, False),
    (2, 2, True, False, True, False), (3, 2, True, False, True, False), (2, 3, True, False, True, False),
    (2, 1, True, False, True, True), (3, 2, True, False, True, True), (2, 2, True, False, True, True), (3, 3, True, False, True, True)])
def test_dist_pagerank_gloo(tmp_path, world, chunks, push, broken, pipelined, packed):
    """chunks > 1: the sweep is enqueued in row chunks and each chunk's piece is all-gathered (async) while
    the next chunk is computed -- the overlap path the GPU ranks take for N > 1.
    push: the exchange by direct copies into the peers' replicas (files here, hipIpc-mapped HBM on the GPUs),
    ordered by the per-step all-reduce of diff.  broken: the copies do nothing -- the first exchange is checked
    against a collective one and every rank falls back to the all-gather.
    pipelined: an engine whose step reads the peers' contributions in two gather calls (hub pieces / the rest),
    driven in DistPageRank's pipelined order -- gather(0) right after the per-step barrier, gather(1) after the early
    barrier of the previous step's tail chunk.
    packed: only the positions a peer reads travel, through landing zones, and are scattered by unpack() behind the
    barriers (DistPageRank's packed order); everything else in the replicas is NaN."""
    scale = 11
    mp.spawn(_worker, args=(world, _free_port(), scale, str(tmp_path), chunks, push, broken, pipelined, packed), nprocs=world, join=True)
    g = po.rmat_graph(scale, permute=True)
    want, it, want_diff = po.pagerank(g, 0.001, 0.85, 100, nthreads=1)
    got = np.zeros(g.N)
    for r in range(world):
        cnt, diff, lo, hi = np.load(tmp_path / ("meta%d.npy" % r))
        assert int(cnt) == it
        assert abs(diff - want_diff) < 1e-11
        got[int(lo):int(hi)] = np.load(tmp_path / ("rank%d.npy" % r))
    assert np.max(np.abs(got - want) / want) < 1e-12


def test_single_process_world1():
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "green-marl_amd"))
    from dist_pagerank import DistPageRank
    g = po.rmat_graph(10)
    pr = DistPageRank(NumpyEngine(g, 0, 1))
    cnt, _ = pr.run(0.001, 0.85, 100)
    want, it, _ = po.pagerank(g, 0.001, 0.85, 100, nthreads=1)
    assert cnt == it
    assert np.max(np.abs(pr.engine.rank_v - want) / want) < 1e-12
