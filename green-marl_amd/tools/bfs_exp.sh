#!/bin/bash
# development helper: per-launch kernel times of the last of 3 hop_dist traversals (RMAT-26, root 0) for each value of an
# environment knob.   usage: bfs_exp.sh VAR v1 v2 ...     (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
var=$1; shift
cat > /tmp/bfs_one.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "green-marl_amd"))
import gmx
g = gmx.Graph.rmat(1 << 26, 16 << 26, 1997, 0.57, 0.19, 0.19, False)
for _ in range(3):
    dist, s = g.hop_dist(0)
print("hop_dist %.3f ms levels %d reached %d examined %d" % (s["kernel_ms"], s["iterations"], s["vertices_reached"], s["edges_examined"]))
PY
for v in "$@"; do
  export $var=$v
  d=gpurun_out/bfsexp_${var}_$(echo "$v" | tr '/.' '__')
  rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -- python3 /tmp/bfs_one.py > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== $var=$v: $(grep hop_dist $d.log)"
  BFS_TRACE_DIR=$d python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["BFS_TRACE_DIR"] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "bfs_" in r["Kernel_Name"]]
last = max(i for i, r in enumerate(rows) if "bfs_init_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%9.1f us  +%8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0]))
PY
done
