# step time of rank 0 of an N-rank partition of RMAT-26 for several values of one environment knob
# usage: ranks_sweep.sh "<ranks...>" VAR "<values...>"
cd $GRAFT_REPO_ROOT
for n in $1; do
  for v in $3; do
    echo "ranks $n $2=$v"
    env SWEEP_RANKS=$n SWEEP_STEPS=10 $2=$v python3 green-marl_amd/tools/cold_sweep.py 26 4 -2 2>&1 | grep "ms/step" || exit 1
  done
done
