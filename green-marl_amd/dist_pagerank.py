"""Multi-GPU PageRank driver: 1-D vertex partition, one process per GPU, RCCL over xGMI.

Mirrors the control flow of the emitted `pagerank` (/root/reference/apps/src/pagerank.gm:9-19):
    Do { diff = 0; <one sweep over the nodes>; cnt++; } While ((diff > e) && (cnt < max));
with the sweep split over ranks:
  * rank k owns an equal-sized, contiguous range of the internally renumbered vertices and a full
    replica of the contribution vector;
  * after every sweep the owned ranges are exchanged with ONE in-place all-gather
    (torch.distributed `all_gather_into_tensor`, backend "nccl" = RCCL): on the fully connected
    xGMI topology every GPU sends V*s/G bytes to each peer directly, which is the
    bandwidth-optimal form of the north star's "all-reduce of the rank vector" (owned range
    filled, rest zero, sum = concatenation);
  * with C > 1 row chunks (engine.set_chunks) the sweep is enqueued chunk by chunk and the all-gather of
    chunk c's freshly written piece runs (async, on the process group's own stream) while chunk c+1 is
    computed -- the exchange hides behind the sweep instead of following it;
  * exchange="push" replaces the all-gather by direct copies: the ranks map each other's replicas (hipIpc)
    and every rank copies each finished chunk into all peers' replicas with the copy engines over the
    point-to-point xGMI links (gmx_pr_push_*), which needs no CUs and therefore really runs under the
    persistent sweep kernel; the per-step barrier that orders the ranks is the all-reduce of `diff`;
  * pushed + pipelined (two row chunks and an engine whose step splits its gather phase, GmxEngine on a plan with
    every in-edge binned): the exchange of the long tail chunk (most of the bytes) runs under the NEXT step's
    gather over the hub tiles (most of the work), which needs only the small hub chunk -- see _step_pipelined;
  * `diff` is a 1-element fp64 all-reduce(SUM) -- the ATOMIC_ADD<double>(&diff, diff_prv) of the
    emitted code with ranks in place of threads.

The engine is anything with the gmx.PageRankState stepping interface (step / contrib tensors /
diff tensor).  The product engine is GmxEngine (HIP kernels through libgmx.so); the CPU tests
drive the same orchestration over gloo with a test-owned engine.
"""
import torch
import torch.distributed as dist


class GmxEngine:
    """gmx.PageRankState behind the engine interface, tensors wrapping the library's HBM buffers."""

    def __init__(self, gmx, graph, elem_bytes, rank, nranks, options):
        self.gmx = gmx
        self.state = gmx.PageRankState(graph, elem_bytes, rank, nranks, options)
        self._cache = {}

    def _wrap(self, dev_array):
        cai = dev_array.__cuda_array_interface__
        key = (cai["data"][0], cai["shape"][0], cai["typestr"])
        t = self._cache.get(key)
        if t is None:
            t = torch.as_tensor(dev_array, device="cuda")
            self._cache[key] = t
        return t

    def reset(self, d):
        self.state.reset(d)

    def step(self):
        self.state.step(None)   # default stream == torch's current stream: ordered with the collectives

    def set_chunks(self, chunks):
        return self.state.set_chunks(chunks)

    def num_chunks(self):
        return self.state.num_chunks()

    def chunk_range(self, chunk):
        return self.state.chunk_range(chunk)

    def step_chunk(self, chunk):
        self.state.step_chunk(chunk, None)

    def contrib_next_full(self):
        return self._wrap(self.state.contrib_next_full())

    # peer push
    def ipc_handles(self):
        return self.state.ipc_handles()

    def set_peers(self, handles):
        self.state.set_peers(handles)

    def push_chunk(self, chunk):
        self.state.push_chunk(chunk, None)

    def push_current(self):
        self.state.push_current(None)

    def push_join(self):
        self.state.push_join(None)

    def unpack(self, chunk=-1):
        self.state.unpack(chunk, None)

    def packed(self):
        return bool(getattr(self.state, "packed_push", False))

    def recv_list(self, r):
        return self._wrap(self.state.recv_list(r))

    def exchange_bytes(self):
        return self.state.exchange_bytes()

    def push_join_chunk(self, chunk, stream=None):
        self.state.push_join_chunk(chunk, stream)

    def gather_classes(self):
        return self.state.gather_classes()

    def step_gather(self, tile_class):
        self.state.step_gather(tile_class, None)

    def contrib_slice(self):
        return self._wrap(self.state.contrib_slice())

    def contrib_full(self):
        return self._wrap(self.state.contrib_full())

    def diff_tensor(self):
        return self._wrap(self.state.diff_dev())

    def exchange_count(self):
        return self.state.exchange_count()

    def download(self, out=None):
        return self.state.download(out)


class DistPageRank:
    """exchange: "collective" (all-gather), "push" (peer copies, engine must offer ipc_handles/set_peers/push_*),
    or "auto" (push if every rank could map its peers, else collective).
    barrier: how ranks are ordered after a pushed step -- "collective" (all-reduce of diff on the device, the
    production path) or "host" (stream sync + host barrier; for ranks sharing one GPU, where RCCL cannot run)."""

    def __init__(self, engine, group=None, always_exchange=False, exchange="collective", barrier="collective",
                 verify_push=True, pipeline=True):
        self.engine = engine
        self.group = group
        self.initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.initialized else 1
        # always_exchange: issue the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.always_exchange = always_exchange and self.initialized
        self.cnt = 0
        self.barrier = barrier
        self.verify_push = verify_push    # check the first pushed exchange against a collective one
        self.push_verified = False
        self._diff = None
        self.exchange = "collective"
        if exchange in ("push", "auto") and self.world > 1:
            self.exchange = "push" if self._setup_push(exchange == "push") else "collective"
        # pipelined pushed step: its second barrier (tail chunk landed everywhere) runs beside the per-step one, so
        # it gets a process group -- i.e. a communicator and stream -- of its own.  Every rank decides the same way.
        self.pipeline = pipeline
        self._early_group = None
        self._early_work = None
        self._early_token = None
        self._side = None
        if pipeline and self.exchange == "push":
            mine = int(hasattr(engine, "gather_classes") and engine.gather_classes() == 2)
            flags = [None] * self.world
            dist.all_gather_object(flags, mine, group=self.group)
            if all(flags):
                self._early_group = dist.new_group(backend=dist.get_backend(group))

    def _setup_push(self, required):
        """Pass the replica handles around and map the peers'; every rank must succeed or none uses it."""
        ok, err = 1, None
        try:
            mine = self.engine.ipc_handles()
        except Exception as e:   # noqa: BLE001 -- any failure means "no push here", reported below
            mine, ok, err = None, 0, e
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=self.group)
        if ok and all(h is not None for h in handles):
            try:
                self.engine.set_peers(handles)
            except Exception as e:   # noqa: BLE001
                ok, err = 0, e
        else:
            ok = 0
        flags = [None] * self.world
        dist.all_gather_object(flags, ok, group=self.group)
        if all(flags):
            return True
        if required:
            raise RuntimeError("peer push unavailable on rank(s) %s: %s" % ([r for r, f in enumerate(flags) if not f], err))
        return False

    def _packed(self):
        return hasattr(self.engine, "packed") and self.engine.packed()

    def _unpack(self, chunk=-1):
        if self._packed():
            self.engine.unpack(chunk)

    def _push_matches_collective(self):
        """The replica of this rank against an all-gather of the owned prefixes; with the packed push only on the
        positions this rank reads (the others are never sent)."""
        full = self.engine.contrib_full()
        mine = self.engine.contrib_slice()
        n = mine.numel()
        need = self.engine.exchange_count() if hasattr(self.engine, "exchange_count") else n
        rank = dist.get_rank(self.group)
        host = self.barrier == "host"       # ranks sharing one GPU: the host barrier already ordered the copies
        got = [torch.empty_like(mine[:need]).cpu() if host else torch.empty_like(mine[:need]) for _ in range(self.world)]
        dist.all_gather(got, mine[:need].cpu() if host else mine[:need].clone(), group=self.group)
        for r in range(self.world):
            have = full[r * n:r * n + need].cpu() if host else full[r * n:r * n + need]
            if self._packed() and r != rank:
                idx = self.engine.recv_list(r).long()
                idx = idx.cpu() if host else idx
                if not bool(torch.equal(got[r][idx], have[idx])):
                    return False
            elif not bool(torch.equal(got[r], have)):
                return False
        return True

    def _rank_barrier(self):
        """After a pushed step: nobody goes on before every rank's copies have completed.  Carries diff."""
        t = self.engine.diff_tensor().clone()
        if self.barrier == "host":
            import torch
            if t.is_cuda:
                torch.cuda.synchronize()
            t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self._diff = t

    def _can_pipeline(self, chunks):
        return self._early_group is not None and chunks == 2 and (self.push_verified or not self.verify_push)

    def _early_barrier(self, chunk):
        """Issued right after chunk `chunk` (every one but the last) has been pushed: completes once every rank's
        copies of it have landed.  Does not hold up this stream; _wait_early() is where the result is needed."""
        eng = self.engine
        if self.barrier == "host":
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            dist.barrier(group=self._early_group)
            return
        if self._side is None:
            self._side = torch.cuda.Stream()
            self._early_token = torch.zeros(1, device="cuda")
            self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            eng.push_join_chunk(chunk, self._side.cuda_stream)     # the side stream waits for the copies ...
            # ... and the collective (on the second group's own stream) for the side stream
            self._early_work = dist.all_reduce(self._early_token, group=self._early_group, async_op=True)

    def _wait_early(self):
        if self._early_work is not None:
            self._early_work.wait()        # orders the current stream after the collective; the host goes on
            self._early_work = None

    def _step_pipelined(self):
        """Two chunks: 0 = tail of the rank's range (most rows, a third of the edges, most of the BYTES),
        1 = hub (most of the work, few bytes).  Only the gather phase of a step reads the peers' contributions:
          gather(0)  hub tiles    needs every rank's hub chunk of the previous step  -> that step's diff all-reduce
          gather(1)  other tiles  needs the tail chunks too                          -> the early barrier
        so the tail chunk travels under gather(0) of the next step.  Replica reuse (double buffered) is safe: a
        rank pushes into a replica only after a barrier that every rank passed after its last read of it."""
        eng = self.engine
        eng.step_gather(0)
        self._wait_early()
        self._unpack(0)                    # (packed push) the tail chunk of the previous step, landed by now
        self._tail_pending = False
        eng.step_gather(1)
        eng.step_chunk(0)
        eng.push_chunk(0)
        self._early_barrier(0)
        self._tail_pending = True          # pushed, not yet unpacked: the next step's start or drain() does that
        eng.step_chunk(1)
        eng.push_chunk(1)
        if self.barrier == "host":
            eng.push_join()
        else:
            eng.push_join_chunk(1)         # this stream waits for the hub copies only
        self._rank_barrier()
        self._unpack(1)                    # (packed push) the hub chunk, which the next step's first gather reads

    def drain(self):
        """Everything issued so far (incl. the travelling tail chunk) ordered before what this stream does next."""
        if self._early_work is not None:
            self._wait_early()
        if getattr(self, "_tail_pending", False):   # (with the host barrier there is no pending collective to tell)
            self._unpack(0)
            self._tail_pending = False

    def _step_pushed(self, chunks):
        eng = self.engine
        if self._can_pipeline(chunks):
            return self._step_pipelined()
        for c in range(chunks):
            eng.step_chunk(c)
            eng.push_chunk(c)      # copies start when the chunk is done, the stream goes on with the next one
        eng.push_join()
        self._rank_barrier()
        self._unpack(-1)

    def _exchange(self):
        if self.world == 1 and not self.always_exchange:
            return
        if self.exchange == "push":
            ok = 1
            try:
                self.engine.push_current()
                self.engine.push_join()
            except Exception:   # noqa: BLE001 -- a refused peer copy: every rank falls back together below
                if not (self.verify_push and not self.push_verified):
                    raise
                ok = 0
            self._rank_barrier()
            if ok:
                self._unpack(-1)
            if self.verify_push and not self.push_verified:
                # once, outside any timed region: what the peers pushed into this replica must be what a
                # collective all-gather of the same slices delivers; otherwise every rank falls back to it
                self.push_verified = True
                ok = ok and self._push_matches_collective()
                flags = [None] * self.world
                dist.all_gather_object(flags, int(ok), group=self.group)
                if not all(flags):
                    self.exchange = "collective"
                    self._exchange()
            return
        full = self.engine.contrib_full()
        mine = self.engine.contrib_slice()
        n = mine.numel()
        need = self.engine.exchange_count() if hasattr(self.engine, "exchange_count") else n
        backend = dist.get_backend(self.group)
        if need < n:
            # only the leading `need` entries of every rank's range are ever read by others (the rest are
            # vertices without out-edges): gather those prefixes straight into their places in the replica
            parts = [full[r * n:r * n + need] for r in range(self.world)]
            dist.all_gather(parts, mine[:need].clone(), group=self.group)
        elif backend == "gloo":
            parts = list(full.chunk(self.world))
            dist.all_gather(parts, mine.clone(), group=self.group)
        else:
            dist.all_gather_into_tensor(full, mine, group=self.group)

    def reset(self, d=0.85):
        self.drain()
        self.engine.reset(d)
        self.cnt = 0
        self._exchange()

    def _step_chunked(self, chunks):
        """Sweep chunk c+1 while chunk c's new contributions travel."""
        eng = self.engine
        n = eng.contrib_slice().numel()
        nxt = eng.contrib_next_full()          # the replica this step writes
        rank = dist.get_rank(self.group)
        pending = []
        for c in range(chunks):
            eng.step_chunk(c)
            off, cnt = eng.chunk_range(c)
            if cnt == 0:
                continue
            parts = [nxt[r * n + off:r * n + off + cnt] for r in range(self.world)]
            # async: the collective is ordered after the kernels enqueued so far and runs on the process
            # group's stream; this stream goes on with the next chunk
            pending.append(dist.all_gather(parts, parts[rank].clone(), group=self.group, async_op=True))
        for w in pending:
            w.wait()                           # the next sweep reads the whole replica

    def step(self):
        """One PageRank iteration of the whole job (local sweep + exchange); asynchronous on GPU."""
        chunks = self.engine.num_chunks() if hasattr(self.engine, "num_chunks") else 1
        self._diff = None
        if self.exchange == "push":
            self._step_pushed(chunks)
        elif chunks > 1 and (self.world > 1 or self.always_exchange):
            self._step_chunked(chunks)
        else:
            self.engine.step()
            self._exchange()
        self.cnt += 1

    def diff(self):
        if self._diff is not None:      # the pushed step's barrier already summed it
            return float(self._diff.item())
        t = self.engine.diff_tensor()
        if self.world > 1 or self.always_exchange:
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def run(self, e=0.001, d=0.85, max_iter=100):
        self.reset(d)
        while True:
            self.step()
            diff = self.diff()
            if not (diff > e and self.cnt < max_iter):
                break
        return self.cnt, diff
