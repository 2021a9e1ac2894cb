# 3-stream trace of the bench step + the exposed-kernel analysis + the per-shape table of the instrumented step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c
mkdir -p $O
B="--no-cpu-baseline --no-extra"
DYNMM_BENCH_SHAPES=$O/shapes.txt python $R/bench.py --steps 5 --warmup 2 $B > $O/bench_shapes.log 2>&1
rocprofv3 --kernel-trace -d $O/multi -o bench -- python $R/bench.py --steps 3 --warmup 1 $B --no-kernel-timing > $O/multi_stdout.log 2>&1
cd $R
python profiles/exposed_kernels.py $O/multi/bench_results.db > $O/exposed.txt 2>&1
rm -rf $O/multi
head -60 $O/exposed.txt
