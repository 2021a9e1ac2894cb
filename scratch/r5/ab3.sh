cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "stem or priority" 2>&1 | tail -5 | tee $O/stem_tests2.log
timeout 1200 python -m pytest tests/test_hip_model.py tests/test_engine.py -x -q -m gpu 2>&1 | tail -5 | tee $O/model_tests2.log
B="--steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-kernel-timing"
for rep in 1 2 3; do
  v=$(timeout 300 python bench.py $B 2>$O/ab3_err.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "new : $v ms" | tee -a $O/ab3.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/multi3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing > $GRAFT_REPO_ROOT/$O/multi3_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/exposed_kernels.py $O/multi3/bench_results.db > $O/exposed3.txt 2>&1
rm -rf $O/multi3
head -12 $O/exposed3.txt; tail -25 $O/exposed3.txt
