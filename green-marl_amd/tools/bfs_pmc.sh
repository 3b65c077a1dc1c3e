#!/bin/bash
# development helper: PMC passes over 3 hop_dist traversals of RMAT-26 (root 0); prints the counters of the
# bfs_bottomup_part_kernel launches of the last traversal.   usage: bfs_pmc.sh <tag>   (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-bfspmc}
cat > /tmp/bfs_one.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "green-marl_amd"))
import gmx
g = gmx.Graph.rmat(1 << 26, 16 << 26, 1997, 0.57, 0.19, 0.19, False)
for _ in range(3):
    dist, s = g.hop_dist(0)
print("hop_dist %.3f ms levels %d reached %d examined %d" % (s["kernel_ms"], s["iterations"], s["vertices_reached"], s["edges_examined"]))
PY
i=0
for set in "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  d=gpurun_out/${tag}_p$i
  rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -- python3 /tmp/bfs_one.py > $d.log 2>&1 || { echo "pass $i ($set) failed"; grep -m2 "error code\|Could not" $d.log; continue; }
  PMC_DIR=$d python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["PMC_DIR"] + "/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "bottomup" in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in by.items():
    print("%-44s %s" % (k, "  ".join("%14.0f" % x for x in v[-3:])))
PY
done
