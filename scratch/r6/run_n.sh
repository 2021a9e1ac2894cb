#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing --config S"
for rep in 1 2 3 4 5 6; do
for cfg in - scratch/r6/libdynmm_w2dring2.so; do
  v=$(timeout 300 python scratch/r5/ab_lib.py $cfg $B 2>$O/ab_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "S $cfg : $v ms" | tee -a $O/ab.log
done
done
tail -3 $O/ab_err.log
