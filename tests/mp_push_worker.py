"""One rank of the peer-push exchange test (tests/test_gpu_parity.py starts WORLD_SIZE of these, all on
cuda:0 -- hipIpc works between processes sharing a device, RCCL does not, so the per-step barrier is the host
one).  Product code under test: gmx_ipc_* / gmx_pr_set_peers / gmx_pr_push_* and DistPageRank(exchange="push").
Rank 0 checks the assembled ranks against the oracle and the exit code reports it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(HERE, "..", "green-marl_amd"), os.path.join(HERE, "..", "oracle")):
    sys.path.insert(0, os.path.abspath(p))


def main():
    scale, chunks, elem, iters = (int(a) for a in sys.argv[1:5])
    binned = len(sys.argv) > 5 and sys.argv[5] == "binned"     # every in-edge binned: the pipelined pushed step
    if binned:
        os.environ["GMX_PR_COLD"] = "0"
    import torch
    import torch.distributed as dist
    import gmx
    import pyoracle as po
    from dist_pagerank import DistPageRank, GmxEngine

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gmx.require_device()
    og = po.rmat_graph(scale, permute=True)
    g = gmx.Graph.upload(og.begin, og.node_idx, og.r_begin, og.r_node_idx)
    options = gmx.GMX_PR_RELABEL | gmx.GMX_PR_HOT_LDS | gmx.GMX_PR_SLICED | (gmx.GMX_PR_COLD_PB if binned else 0)
    eng = GmxEngine(gmx, g, elem, rank, world, options)
    assert eng.set_chunks(chunks) == chunks
    pr = DistPageRank(eng, exchange="push", barrier="host")
    assert pr.exchange == "push"
    assert eng.packed() == (os.environ.get("GMX_PUSH_PACKED", "1") != "0")     # the packed push unless switched off
    piped = []
    if binned:
        assert eng.gather_classes() == 2 and pr._early_group is not None
        orig = pr._step_pipelined
        pr._step_pipelined = lambda: (piped.append(1), orig())[1]
    cnt, diff = pr.run(1e-300, 0.85, iters)
    assert len(piped) == (cnt if binned and chunks == 2 else 0)
    out = np.zeros(og.N, dtype=np.float32 if elem == 4 else np.float64)
    eng.download(out)                      # fills the vertices this rank owns
    t = torch.from_numpy(out.astype(np.float64))
    dist.all_reduce(t)
    ok = True
    if rank == 0:
        want, it, want_diff = po.pagerank(og, 1e-300, 0.85, iters)
        err = float(np.max(np.abs(t.numpy() - want) / want))
        tol = 1e-6 if elem == 4 else 1e-12
        ok = cnt == it and err < tol and abs(diff - want_diff) <= (1e-3 if elem == 4 else 1e-9) * want_diff
        print("push exchange: world=%d chunks=%d elem=%d iters=%d rel_err=%.3e diff=%.6e want_diff=%.6e -> %s"
              % (world, chunks, elem, cnt, err, diff, want_diff, "OK" if ok else "MISMATCH"), flush=True)
    dist.barrier()
    eng.state.free()
    g.free()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
