#!/bin/bash
# alternating runs of the whole step over several builds of libdynmm_hip.so: $1 = tag, the rest = libraries ("-" = the product)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4 5; do
for cfg in "$@"; do
  v=$(timeout 300 python scratch/r5/ab_lib.py $cfg $B 2>$O/ab_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/ab.log
done
done
tail -3 $O/ab_err.log
