#!/usr/bin/env python3
"""bench.py -- headline benchmark: edges/s (GTEPS) per PageRank iteration on RMAT-26.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one PageRank iteration (pagerank.gm:9-19 loop body) of the WHOLE job: the local
neighbour-reduction sweep of every rank plus, for N > 1, the all-gather of the new contribution
slices.  Inputs (CSR, rank and contribution vectors) are resident in HBM before the timed region.
The graph is the reference's own RMAT generator (graph_gen.cc:159-287, seed 1997, a,b,c =
.57,.19,.19, edge factor 16, permute=true) run on the device, then do_semi_sort +
make_reverse_edges exactly as load_binary does.  Total work is fixed as N grows ("strong").

Rank 0 prints ONE JSON line with the driver's contract fields plus `roofline`, `cpu_baseline` and, at N = 1,
`extra`: the other BASELINE.json configurations measured in the same run (PageRank fp64 on the same graph,
PageRank fp32 on RMAT-24, hop_dist from vertex 0 on RMAT-26, triangle counting on symmetrised RMAT-24).
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

# the CPU baseline pins its OpenMP threads like the reference's run.sh does (scripts/run.sh:237-238,350);
# libgomp reads this once, when it is first loaded
os.environ.setdefault("GOMP_CPU_AFFINITY", "0-%d" % ((os.cpu_count() or 1) - 1))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "green-marl_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


# the sources of the kernels of one PageRank step (the bench line's roofline kernels) and of the plan they run on
STEP_SOURCES = ("gmx_pagerank.hip", "gmx_pr_cold.hip")


def kernel_code_hash():
    """sha256 over the HIP sources of the PageRank step: a PMC traffic figure is only quoted for the code it was measured
    on (the other translation units -- traversal, triangle counting, graph construction -- do not run in the timed step)."""
    h = hashlib.sha256()
    for name in STEP_SOURCES:
        f = os.path.join(ROOT, "green-marl_amd", "csrc", name)
        h.update(name.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(gmx, graph, scale):
    """The oracle's OpenMP restatement of the emitted pagerank (kind "port") on the SAME graph the GPU line is
    measured on, on this box's host cores: one warm-up iteration, then the median of three single iterations
    (BASELINE.md section 2).  The checker is only TIMED here; nothing it computes feeds the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    begin, node_idx, rb, rn = graph.download()
    og = po.Graph(graph.V, begin, node_idx, rb, rn)
    cores = po.lib().gmo_max_threads()
    po.pagerank(og, 1e-300, 0.85, 1, nthreads=cores)     # touch pages / warm up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        po.pagerank(og, 1e-300, 0.85, 1, nthreads=cores)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return {"value": og.M / dt / 1e9, "unit": "GTEPS", "cores": cores, "kind": "port",
            "sample": "the bench graph itself (RMAT-%d, E=%d): 1 warm-up + 3 single iterations of the fp64 OpenMP "
                      "restatement of the emitted pagerank loop (schedule(dynamic,128)), median %.2f s per iteration "
                      "(%.2f / %.2f / %.2f), GOMP_CPU_AFFINITY=%s" % (scale, og.M, dt, times[0], times[1], times[2],
                                                                   os.environ.get("GOMP_CPU_AFFINITY", ""))}


def pagerank_steps(gmx, graph, elem, steps, warmup):
    """ms per iteration of the stepping API on one GPU (hipEvents around every step's kernels)."""
    st = gmx.PageRankState(graph, elem, 0, 1, gmx.default_pr_options(graph.V, 1))
    st.reset(0.85)
    for _ in range(warmup):
        st.step()
    st.timing(True)
    for _ in range(steps):
        st.step()
    n, ms = st.kernel_time()
    work = st.work()
    st.free()
    return {"ms_per_iter": ms, "gteps": graph.E / (ms * 1e-3) / 1e9,
            "roofline_frac": work["algorithmic_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "iterations_timed": n}


def extras(gmx, graph26, scale):
    """The other BASELINE.json configurations, measured in this run (1 GPU)."""
    out = {}
    out["pagerank_f64_rmat%d" % scale] = dict(pagerank_steps(gmx, graph26, 8, 10, 3),
                                               note="same graph as the headline, Node_Prop<Double> (the reference's signature); "
                                                    "algorithmic bytes 12E + 32V")
    g24 = gmx.Graph.rmat(1 << 24, 16 << 24, 1997, 0.57, 0.19, 0.19, True)
    out["pagerank_f32_rmat24"] = dict(pagerank_steps(gmx, g24, 4, 20, 3), note="BASELINE configs[1]; algorithmic bytes 8E + 20V")
    # the drop-in call: what the reference's driver times as `running time` (common_main.h:198-201) wraps the whole
    # pagerank() entry -- plan build on the first call of a graph, iterations until diff <= e, rank download
    for name, gr in (("pagerank_entry_rmat24", g24), ("pagerank_entry_rmat%d" % scale, graph26)):
        import numpy as np
        calls = []
        for _ in range(2):
            t0 = time.perf_counter()
            _, st = gr.pagerank(0.001, 0.85, 100, np.float64)
            calls.append((time.perf_counter() - t0) * 1e3)
        out[name] = {"first_call_ms": calls[0], "cached_call_ms": calls[1], "iterations": st["iterations"], "iterations_ms": st["kernel_ms"],
                     "download_ms": st["d2h_ms"],
                     "note": "gmx_pagerank_f64(e=0.001, d=0.85, max=100) wall clock: first call builds and caches the plan, "
                             "the second reuses it; both include the iterations and the copy of rank[] to the host"}
    # the SURVEY 8f rank 3-4 apps on the same RMAT-24 (device time of the second call; inputs as the reference's drivers draw
    # them in spirit: random edge lengths 1..100, random ages 0..99, four random groups, the top hub and four more seeds)
    rng = np.random.default_rng(1)
    hub = int(np.argmax(np.diff(g24.download(reverse=False)[0])))
    length, age, member = rng.integers(1, 101, g24.E).astype(np.int32), rng.integers(0, 100, g24.V).astype(np.int32), rng.integers(0, 4, g24.V).astype(np.int32)
    apps = {}
    for _ in range(2):
        apps["sssp_ms"] = g24.sssp(length, hub)[1]["kernel_ms"]
        apps["avg_teen_cnt_ms"] = g24.avg_teen_cnt(age, 5)[2]["kernel_ms"]
        apps["conduct_ms"] = g24.conduct(member, 1)[1]["kernel_ms"]
        apps["bc_5_seeds_ms"] = g24.bc(np.array([hub, 1, 2, 3, 4], np.int32))[1]["kernel_ms"]
    del length, age, member
    out["apps_rmat24"] = dict(apps, note="sssp from the top hub, avg_teen_cnt(K=5), conduct(group 1), comp_BC(hub + 4 seeds): device time, "
                                           "results checked against the oracle in tests/ (not here)")
    # triangle counting on the symmetrised simple version of the same RMAT-24
    gs = g24.symmetrize()
    g24.free()
    t0 = time.perf_counter()
    T, st = gs.triangle_counting()
    tc_first_wall = time.perf_counter() - t0    # incl. the symmetry check and the degree-ordered copy with its hub matrix
    t0 = time.perf_counter()
    T2, st2 = gs.triangle_counting()            # second call: the degree-ordered copy is cached, like the reverse CSR
    tc_warm_wall = time.perf_counter() - t0
    import numpy as np
    deg = np.diff(gs.download(reverse=False)[0]).astype(np.float64)
    merge_bytes = 4.0 * float(np.sum(deg * deg))    # sum over edges (v,u), u > v of 4 (d(v) + d(u)) = 4 sum_v d(v)^2 on a symmetric simple graph
    out["triangle_counting_rmat24_sym"] = {"seconds": st2["kernel_ms"] * 1e-3, "first_call_seconds": st["kernel_ms"] * 1e-3,
                                           "first_call_wall_seconds": tc_first_wall, "warm_call_wall_seconds": tc_warm_wall,
                                            "triangles": T, "edges": gs.E, "gteps": gs.E / (st2["kernel_ms"] * 1e-3) / 1e9,
                                            "merge_form_algorithmic_bytes": merge_bytes,
                                            "merge_form_gbs": merge_bytes / (st2["kernel_ms"] * 1e-3) / 1e9,
                                            "note": "BASELINE configs[4]; E = edge slots of the symmetrised graph; merge_form_* = SURVEY 8d's figure for the "
                                                    "sorted-merge form in the emitted vertex order (4 sum d^2) over the measured time -- the degree-ordered "
                                                    "kernel skips most of that work; its measured HBM traffic and unit utilisation: "
                                                    "profiles/round3_d_tc_rmat24_pmc.txt"}
    assert T == T2
    gs.free()
    # hop_dist from vertex 0 on RMAT-26 without the final permutation (vertex 0 is then the top hub; with
    # permute=true vertex 0 may be isolated)
    gb = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)
    runs, walls = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        _, st = gb.hop_dist(0)
        walls.append((time.perf_counter() - t0) * 1e3)
        runs.append(st)
    st = sorted(runs[1:], key=lambda r: r["kernel_ms"])[1]
    t = st["kernel_ms"] * 1e-3
    out["hop_dist_rmat%d_root0" % scale] = {
        "ms": st["kernel_ms"], "levels": st["iterations"], "vertices_reached": st["vertices_reached"],
        "edges_reached": st["edges_reached"], "edges_examined": st["edges_examined"],
        "gteps": st["edges_reached"] / t / 1e9,
        "first_call_wall_ms": walls[0], "warm_call_wall_ms": sorted(walls[1:])[1], "download_ms": st["d2h_ms"],
        "roofline_frac": (8 * st["edges_reached"] + 12 * st["vertices_reached"]) / t / 1e9 / HBM_PEAK_GBS,
        "note": "BASELINE configs[2]; ms = median of 3 warm traversals (device time of the traversal); the wall-clock figures are whole "
                "gmx_hop_dist calls incl. the 256 MB dist[] download, the first one also the per-graph bottom-up hint and traversal "
                "state; TEPS in the Graph500 convention (out-edges of the reached vertices / time); algorithmic bytes 8 E_r + 12 V_r -- "
                "direction optimisation examines fewer edges, so this fraction is a work-skipping figure, not bandwidth (measured "
                "traffic: profiles/round3_d_bfs_rmat26_pmc.txt)"}
    gb.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--edge-factor", type=int, default=16, help="edges per vertex (22 at scale 26 ~ Twitter-2010's 1.47 G edges)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--options", type=int, default=-1, help="gmx_pr_create option bits (default: library default)")
    ap.add_argument("--chunks", type=int, default=0, help="row chunks per step for N > 1 (0: 2 when the sliced variant runs)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "push", "collective"],
                    help="N > 1: peer copies over xGMI (hipIpc + copy engines) or RCCL all-gather; auto = push if it sets up")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configurations")
    args = ap.parse_args()

    # Keep stdout clean for the ONE JSON line: libraries (RCCL prints a version banner to stdout) are
    # pointed at stderr for the whole run, the result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run); got %d"
                         % (args.gpus, args.gpus, world))

    import torch
    import torch.distributed as dist

    import gmx
    from dist_pagerank import DistPageRank, GmxEngine

    gmx.require_device()                       # no CPU fallback: fail loudly without the HIP path
    # GMX_BENCH_SHARED_GPU=1 (development, a one-GPU box): every rank on device 0, the host-side process group (gloo) and
    # the host barrier -- hipIpc works between processes sharing a device, RCCL does not.  It rehearses this file's
    # N > 1 code (warm-up agreement, timing, exchange check, fallbacks, the JSON line); its numbers mean nothing.
    shared_gpu = os.environ.get("GMX_BENCH_SHARED_GPU") == "1" and world > 1
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    gmx.set_device(local_rank)
    # GMX_BENCH_FORCE_COLLECTIVES=1 (development): run the RCCL calls with a single rank too
    force_coll = os.environ.get("GMX_BENCH_FORCE_COLLECTIVES") == "1" and "MASTER_PORT" in os.environ
    if world > 1 or force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if shared_gpu else "cuda"   # where this file's own small collectives live

    elem = 4 if args.dtype == "f32" else 8
    options = gmx.default_pr_options(1 << args.scale, world) if args.options < 0 else args.options
    N, M = 1 << args.scale, args.edge_factor << args.scale

    t0 = time.perf_counter()
    graph = gmx.Graph.rmat(N, M, 1997, 0.57, 0.19, 0.19, True)
    engine = GmxEngine(gmx, graph, elem, rank, world, options)
    chunks = 1
    if world > 1 or force_coll:
        # chunked step: the low-degree tail of the rank's range is reduced first and travels while the hubs
        # (most of the sweep) are reduced
        chunks = engine.set_chunks(args.chunks if args.chunks > 0 else 2)
    pr = DistPageRank(engine, always_exchange=force_coll, exchange=args.exchange if world > 1 else "collective",
                      barrier="host" if shared_gpu else "collective",
                      pipeline=os.environ.get("GMX_BENCH_PIPELINE", "1") != "0")
    pr.reset(0.85)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0

    def all_agree(flag):
        """True if `flag` holds on every rank."""
        if world == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    def warm():
        try:
            for _ in range(args.warmup):
                pr.step()
            torch.cuda.synchronize()
            return True
        except Exception as e:   # noqa: BLE001 -- reported; every rank then takes the plain step together
            print("bench.py rank %d: warm-up step failed: %r" % (rank, e), file=sys.stderr, flush=True)
            return False

    def timed():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        engine.state.timing(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pr.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t1
        n, ms = engine.state.kernel_time()
        engine.state.timing(False)
        return d, n, ms

    def check_exchange():
        """Outside the timed region: after the last step every rank's replica must hold exactly what an all-gather of
        the owned slices delivers (the pushed, pipelined exchange is ordered by barriers only -- this is its check)."""
        if not (world > 1 and pr.exchange == "push"):
            return None
        pr.drain()
        torch.cuda.synchronize()
        dist.barrier()
        return all_agree(pr._push_matches_collective())

    # The pipelined pushed step has never run on real multi-GPU hardware from the development boxes (one GPU each): if
    # its warm-up fails on some rank, or the replicas are not what an all-gather delivers after the timed steps, every
    # rank drops to the plain pushed step (exchange exposed) and the measurement is repeated.
    def fail(msg):
        """A run whose ranks disagree is not a measurement: no metric line, non-zero exit on every rank."""
        print("bench.py rank %d: %s" % (rank, msg), file=sys.stderr, flush=True)
        sys.exit(1)

    fell_back = False
    ok = all_agree(warm())
    if not ok and world > 1:
        fell_back = True
        pr._early_group = None
        pr.reset(0.85)
        if not all_agree(warm()):
            fail("the warm-up steps failed again after the fallback to the plain pushed step")
    elif not ok:
        fail("the warm-up steps failed")
    dt, launches, kernel_ms = timed()
    last_diff = pr.diff()
    verdict = check_exchange()
    if verdict is False and pr._early_group is not None:
        fell_back = True
        pr._early_group = None
        pr.reset(0.85)
        if not all_agree(warm()):
            fail("the warm-up steps failed after the fallback to the plain pushed step")
        dt, launches, kernel_ms = timed()
        last_diff = pr.diff()
        verdict = check_exchange()
    if verdict is False:
        fail("exchange check failed: after the timed steps some rank's replica is not what an all-gather of the owned "
             "slices delivers; no metric is reported")
    exchange_check = None if verdict is None else "replicas equal an all-gather of the owned slices on every rank"

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    pipelined = bool(world > 1 and pr.exchange == "push" and pr._can_pipeline(chunks))
    work = engine.state.work()
    gteps = graph.E / (ms_per_step * 1e-3) / 1e9
    achieved = work["algorithmic_bytes"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # HBM bytes per step measured offline with rocprofv3 --pmc on this same command (committed under profiles/):
    # quoted only while the kernel sources are the ones it was measured on
    traffic, traffic_source = None, None
    code_hash = kernel_code_hash()
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "bench_traffic.json")))
        ent = tj.get("scale%d_%s_opt%d_gpus%d" % (args.scale, args.dtype, options, world)) if args.edge_factor == 16 else None
        if isinstance(ent, dict) and ent.get("kernel_code_hash") == code_hash:
            traffic, traffic_source = ent["hbm_bytes_per_step"], ent.get("source")
        elif isinstance(ent, dict):
            traffic_source = "stale: %s was measured at kernel code %s, this is %s" % (ent.get("source"), ent.get("kernel_code_hash"), code_hash)
    except (OSError, ValueError):
        pass

    kernel_name = engine.state.kernel_name()
    cpu = extra = copy_gbs = None
    if rank == 0 and world == 1:
        engine.state.free()   # make room: the extras build their own plans
        copy_gbs = gmx.copy_bandwidth(1 << 30, 10)
        if not args.no_extra and args.edge_factor == 16:
            extra = extras(gmx, graph, args.scale)
        if not args.no_cpu:
            cpu = cpu_baseline(gmx, graph, args.scale)

    if rank == 0:
        out = {
            "metric": "edges/s (GTEPS) per PageRank iter, RMAT-%d" % args.scale + ("" if args.edge_factor == 16 else " x%d" % args.edge_factor),
            "value": gteps, "unit": "GTEPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "pagerank RMAT-%d (V=%d, E=%d), reference RMAT generator seed 1997 permute=1, "
                                   "d=0.85, fixed iterations" % (args.scale, N, M),
                       "partition": "1-D vertex, %d rank(s)" % world,
                       "exchange": "none" if world == 1 and not force_coll else
                                   ("peer copies over xGMI (hipIpc, copy engines), %d row chunk(s), barrier = all-reduce of diff"
                                    if pr.exchange == "push" else "RCCL all-gather of contribution slices, %d row chunk(s)") % chunks
                                   + ("; pipelined: the tail chunk travels under the next step's gather over the hub tiles" if pipelined else ""),
                       "exchange_check": exchange_check,
                       "exchange_bytes_per_rank_and_step": (engine.exchange_bytes() if world > 1 else 0),
                       "exchange_packed": bool(world > 1 and engine.packed()),
                       "step_form": "single rank" if world == 1 and not force_coll else
                                    (("pushed, pipelined" if pipelined else "pushed, plain") if pr.exchange == "push" else "collective")
                                    + (" (fell back from the pipelined form)" if fell_back else ""),
                       "options": options, "setup_s": round(setup_s, 2), "last_diff": last_diff,
                       **({"rehearsal": "GMX_BENCH_SHARED_GPU=1: all ranks on ONE GPU, host-side process group and barrier -- "
                                        "this line exercises the N > 1 code, its numbers are not a measurement"} if shared_gpu else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # the bytes the kernels really move (PMC) over the same time: the honest bandwidth figure next to
                         # the contract's algorithmic one (the binned layout moves fewer bytes than 8E + 20V charges)
                         "traffic_gbs": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic and kernel_ms > 0 else None,
                         "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and kernel_ms > 0 else None,
                         "copy_bw_gbs": copy_gbs, "frac_of_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "kernel_code_hash": code_hash,
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": work["algorithmic_bytes"]},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
