"""Multi-GPU drivers for hop_dist and triangle_counting: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL), every rank holding the whole CSR.

hop_dist: level-synchronous, direction-optimising.  Top-down levels (small frontiers) are run by every rank on
the whole frontier -- nothing to exchange.  Bottom-up levels are partitioned by destination range: a rank
finds parents for the unvisited vertices of its range and contributes its slice of the "found" bitmap; one
in-place all-gather of V/8/N bytes per rank and level (the frontier-bitmap exchange of 8e) gives every rank
the whole next frontier.  Frontier sizes are identical on all ranks, so direction and termination need no
further communication -- the `Exist(n){n.updated}` of hop_dist.gm:29 is the popcount every rank computes.

triangle_counting: the edge slots are dealt to the ranks, each counts its share, one all-reduce(SUM, int64)
-- the ATOMIC_ADD<int64_t>(&T, T_prv) of the emitted code with ranks in place of threads.

The engines are anything with the stepping interface of gmx.BfsState / gmx.Graph.triangle_counting; the CPU
tests drive the same orchestration over gloo with test-owned engines."""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class DistHopDist:
    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = _world(group)
        self.levels = 0
        self.exchanges = 0

    def _exchange(self):
        words, off, n = self.engine.found_bitmap()
        full = words if isinstance(words, torch.Tensor) else torch.as_tensor(words, device="cuda")
        mine = full[off:off + n]
        if dist.get_backend(self.group) == "gloo":
            dist.all_gather(list(full.chunk(self.world)), mine.clone(), group=self.group)
        else:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        self.exchanges += 1

    def run(self, root):
        """Returns the number of levels; the engine then holds dist[] (identical on every rank)."""
        eng = self.engine
        eng.start(root)
        self.levels = self.exchanges = 0
        while True:
            need = eng.step_begin()
            if need and self.world > 1:
                self._exchange()
            if eng.step_end() == 0:
                break
            self.levels += 1
        return self.levels


def dist_triangle_counting(count_part, group=None, device=None):
    """count_part(part, nparts) -> this rank's share (e.g. lambda p, n: graph.triangle_counting(p, n)[0])."""
    world = _world(group)
    rank = dist.get_rank(group) if world > 1 else 0
    t = torch.tensor([int(count_part(rank, world))], dtype=torch.int64, device=device or "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())
