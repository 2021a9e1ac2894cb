cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "conv2d_fwd_bwd or grouped_weight" > $O/s2_tests.log 2>&1
tail -4 $O/s2_tests.log
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
for cfg in scratch/r5/libdynmm_nos2.so -; do
  v=$(timeout 300 python scratch/r5/ab_lib.py $cfg $B 2>$O/ab_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$cfg : $v ms" | tee -a $O/s2_ab.log
done
done
DYNMM_BENCH_SHAPES=$O/shapes.txt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/bench_shapes.log 2>&1
grep -E "wgrad_s2|conv_wgrad<|conv_wgrad_v4" $O/shapes.txt | head -20
