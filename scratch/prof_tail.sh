cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_tail
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extra --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $O/single -o bench -- python $R/bench.py --steps 3 --warmup 1 $B --single-stream > $O/single_stdout.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/single/bench_results.db $O/single.md > /dev/null
rm -rf $O/*/*.db
tail -1 $O/single_stdout.log | cut -c1-200
head -45 $O/single.md | cut -c1-170
python -m pytest tests/test_engine.py tests/test_affect.py -q -m gpu -k "tail or adamw" 2>&1 | tail -3
