#!/bin/bash
mkdir -p gpurun_out/r6f
cd $GRAFT_REPO_ROOT
TAPS=v timeout 600 python scratch/r6/wino_time.py o3 o4 > gpurun_out/r6f/wino_vert_occ.log 2>&1
cat gpurun_out/r6f/wino_vert_occ.log
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for g in "" "--graph"; do
  v=$(timeout 300 python bench.py $B $g 2>gpurun_out/r6f/graph_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('dependent_kernel_interval_us'))" 2>/dev/null)
  echo "graph='$g' : $v" | tee -a gpurun_out/r6f/graph_ab.log
done
done
tail -3 gpurun_out/r6f/graph_err.log
