# per-kernel times of the fp64 binned step (RMAT-26) and per-rank step times of an N-rank partition (fp32)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWEEP_STEPS=6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rk64 -- python3 green-marl_amd/tools/cold_sweep.py 26 8 -2 > gpurun_out/rk64.log 2>&1 || exit 1
grep "ms/step" gpurun_out/rk64.log
python3 green-marl_amd/tools/ktrace.py $(ls -t gpurun_out/rk64/*/*kernel_trace.csv | head -1) pr_cold
for n in 1 2 4 8; do SWEEP_RANKS=$n SWEEP_STEPS=10 python3 green-marl_amd/tools/cold_sweep.py 26 4 -2 2>&1 | grep "ms/step"; done
SWEEP_RANKS=8 SWEEP_CHUNKS=2 SWEEP_STEPS=10 python3 green-marl_amd/tools/cold_sweep.py 26 4 -2 2>&1 | grep "ms/step"
SWEEP_RANKS=2 SWEEP_CHUNKS=2 SWEEP_STEPS=10 python3 green-marl_amd/tools/cold_sweep.py 26 4 -2 2>&1 | grep "ms/step"
