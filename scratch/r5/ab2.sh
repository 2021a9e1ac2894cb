cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "stem" 2>&1 | tail -5 | tee $O/stem_tests.log
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3 4; do
for cfg in "nostem x" "stem x" "stem 1"; do
  set -- $cfg
  v=$(timeout 300 python scratch/r5/ab2.py $1 $2 $B 2>$O/ab2_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$1 wgrad=$2 : $v ms" | tee -a $O/ab2.log
done
done
tail -3 $O/ab2_err.log
