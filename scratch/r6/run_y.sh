#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y; mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do
for n in 2 3; do
  v=$(timeout 300 python scratch/r6/wg3_ab.py $n $B 2>$O/err_$n.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "wgrad streams $n : $v ms   $(grep 'plan report' $O/err_$n.log | cut -c1-400)" | tee -a $O/ab.log
done
done
