cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rank8 -- python3 green-marl_amd/tools/ranks_bench.py 26 4 8 > gpurun_out/prof_rank8.log 2>&1
f=$(find gpurun_out/prof_rank8 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/prof_rank8_kernel_stats.csv
find gpurun_out/prof_rank8 -name "*.csv" ! -name "*kernel_stats.csv" -delete
grep "^\"void pr_cold\|^\"pr_diff\|^\"void pr_pack\|^\"void pr_unpack" gpurun_out/prof_rank8_kernel_stats.csv | cut -c1-70,230-330
grep "^N" gpurun_out/prof_rank8.log | cut -c1-120
