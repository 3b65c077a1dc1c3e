#!/bin/bash
# prof_pr.sh <tag> [scale] [elems] -- rocprofv3 kernel stats of the PageRank stepping loop (tools/dump_plan.py) -> gpurun_out/prof_<tag>*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; scale=${2:-26}; elems=${3:-4,8}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 green-marl_amd/tools/dump_plan.py $scale $elems > gpurun_out/prof_${tag}.log 2>&1
f=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv
find gpurun_out/prof_${tag} -name "*.csv" ! -name "*kernel_stats.csv" -delete
