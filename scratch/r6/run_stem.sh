cd $GRAFT_REPO_ROOT
O=gpurun_out/r6stem; mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4 5; do for cfg in 0 1; do
  v=$(timeout 300 python scratch/r6/stem_ab.py $cfg $B 2>$O/err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "stem dual $cfg : $v ms" | tee -a $O/ab.log
done; done
tail -3 $O/err.log
