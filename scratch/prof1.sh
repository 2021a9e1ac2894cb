mkdir -p gpurun_out/prof_r01
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01/bench_stdout.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r01 | head -30
