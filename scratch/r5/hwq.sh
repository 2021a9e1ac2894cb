cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g
mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for q in 4 8; do
  v=$(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "GPU_MAX_HW_QUEUES=$q : $v ms" | tee -a $O/hwq.log
done
done
