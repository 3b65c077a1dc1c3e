#!/bin/bash
# rocprofv3 passes of the default bench command (GPU leg only): kernel trace + stats, then the PMC passes the
# traffic figure comes from (separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum).
# usage: pmc_bench.sh <tag> [extra bench.py arguments, e.g. --dtype f64]   -> gpurun_out/<tag>_{stats,fetch,write,tcc}/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-bench}
shift
extra="$@"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extra $extra > gpurun_out/${tag}_stats.json 2> gpurun_out/${tag}_stats.err
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extra $extra > /dev/null 2> gpurun_out/${tag}_fetch.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extra $extra > /dev/null 2> gpurun_out/${tag}_write.err
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_tcc -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-extra $extra > /dev/null 2> gpurun_out/${tag}_tcc.err
cat gpurun_out/${tag}_stats.json
