# per-kernel times of rank 0 of an N-rank partition of RMAT-26 (development helper; run through gpurun)
# usage: ranks_trace.sh "8 4 1"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWEEP_STEPS=6
for n in ${1:-8 4 1}; do
  export SWEEP_RANKS=$n
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rk$n -- python3 green-marl_amd/tools/cold_sweep.py 26 4 -2 > gpurun_out/rk$n.log 2>&1 || exit 1
  grep -v "^W2026" gpurun_out/rk$n.log | tail -4
  python3 green-marl_amd/tools/ktrace.py $(ls -t gpurun_out/rk$n/*/*kernel_trace.csv | head -1) pr_cold
done
