#!/bin/bash
# alternating A/B of the whole step between two builds of libdynmm_hip.so: $1 = tag, $2 = alternative library (A), product = B
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in $(seq 1 ${AB_PAIRS:-6}); do
for cfg in $2 -; do
  v=$(timeout 300 python scratch/r5/ab_lib.py $cfg $B 2>$O/ab_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/ab.log
done
done
tail -3 $O/ab_err.log
