# alternating A/B of stream priorities on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for cfg in "x x" "-1 x" "x 1" "-1 1" "1 x"; do
  set -- $cfg
  v=$(timeout 300 python scratch/r5/prio_ab.py $1 $2 $B 2>$O/prio_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "side=$1 wgrad=$2 : $v ms" | tee -a $O/prio_ab.log
done
done
grep -i "priority" $O/prio_err.log | head
