#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do
for cfg in - 1 12 1234; do
  v=$(timeout 300 python scratch/r6/stagger_ab.py $cfg $B 2>$O/err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "stagger $cfg : $v ms" | tee -a $O/stagger_ab.log
done
done
tail -3 $O/err.log
timeout 600 python scratch/r6/w43_time.py v4 v0 v4 v0 > $O/w43_warm.log 2>&1; cat $O/w43_warm.log
