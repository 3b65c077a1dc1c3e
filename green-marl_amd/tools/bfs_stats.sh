# rocprofv3 kernel stats of hop_dist: 5 traversals of RMAT-26 and RMAT-20 from vertex 0 (profiles/round*_bfs_*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/bfs_run.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "green-marl_amd"))
import gmx
scale = int(sys.argv[1])
g = gmx.Graph.rmat(1 << scale, 16 << scale, 1997, 0.57, 0.19, 0.19, False)   # the graph bench.py times (BASELINE configs[2])
for _ in range(5):
    dist, s = g.hop_dist(0)
print("RMAT-%d root 0: %.3f ms, levels %d" % (scale, s["kernel_ms"], s["iterations"]))
PY
for sc in 26 20; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bfs_stats_$sc -- python3 /tmp/bfs_run.py $sc > gpurun_out/bfs_stats_$sc.log 2>&1 || exit 1
  grep RMAT gpurun_out/bfs_stats_$sc.log
done
